// Affine-element owner-computes stiffness kernel, second form (k_affine_rows): interface of the translation unit
// affine_rows.hip.  See that file for the design; engine.hip only builds the tables and launches.
#pragma once
#include <hip/hip_runtime.h>

#include "device_common.hpp"

namespace fenris_hip {

// per-position tables (position = node block all of whose elements are affine, in CSR order)
struct AffineRowTables {
    const int4* hdr;      // [npos]       {r0: first node-level CSR entry, nrow: node-level entries of the block's rows,
                          //               flags (bit 0: every (node, column) block of the rows has an owner lane) | lane table << 8,
                          //               number of slots}
    const uint2* lanes;   // [ntab][256]  lane records, see affine_rows.hip; positions with identical records share a table
    const int* elem;      // [npos][us]   element id per slot (-1: empty)
    const double* rec;    // [E][GW]      element records (R or M) written by affine_records_launch before every launch
    const double* ghat;   // [64][GW]     reference blocks Ghat_ab (all 64 (a, b); LinearElastic GW = 10, Laplace GW = 6)
    int us, npos, acc_max;  // slots per position, positions of this launch, largest S * S * nrow
    int pos0, npos_all;     // first position of this launch (launches may cover a part of the sweep), positions in the tables
    int incomplete;         // some position has a (node, column) block without an owner lane (element masks): staged rows are cleared behind the store
};

constexpr int AFFINE_ROWS_GW_LE = 10, AFFINE_ROWS_GW_LAP = 6;
constexpr int AFFINE_ROWS_NO_CLEAR = 0x40000;   // bits of the launcher's `ablate` argument kept from round 3's timing experiments (scripts/exp_slab_time3.py set them
constexpr int AFFINE_ROWS_NO_CARRY = 0x20000;   // through environment switches that are gone): never set by the library
constexpr int AFFINE_ROWS_PRIO_SHIFT = 24;        // bits 24-25 of the launcher's `ablate` argument: s_setprio level of the store wave, 26-27: of the loader wave
constexpr int AFFINE_ROWS_NT_STORES = 0x10000;  // bit of the launcher's `ablate` argument: non-temporal stores of the rows
constexpr int AFFINE_ROWS_THREADS = 384;  // four row waves + one loader wave + one store wave

size_t affine_rows_lds_bytes(int op, int us, int acc_max);

// lane records, headers and slot vertices of every position from the pipelined kernel's position records (p_rec, layout of
// k_build_pipe_tables) and its per-slot connectivity.  *status (device) is set to 1 when a block cannot be expressed
// (more than 8 terms per block, more than 256 lanes, offsets out of range).
hipError_t affine_rows_build(hipStream_t stream, const int* p_rec, int rw_old, int us, int ms, int nbs, int npos, int S,
                             const unsigned* ncols, const int* p_conn, int cs, const int* p_elem, int4* hdr, uint2* lanes,
                             int* status, unsigned long long* hash, int mirror = 0, unsigned long long* hash2 = nullptr, int max_row = 128);
// (max_row: the longest node row of the pattern -- the builder sizes its key space and its LDS by it; a row longer than 128 makes the position `bad`)
// lanes == nullptr: headers and hashes only (hash2: a second, independent 64-bit hash of the records) -- the records of the positions are
// never written in full; affine_rows_tables then forms the records of the FIRST position of every distinct table once more, straight into
// the compact tables, and puts the table ids into the headers (mismatch[1]: some position has a block without an owner lane)
hipError_t affine_rows_tables(hipStream_t stream, const int* p_rec, int rw_old, int us, int ms, int nbs, int npos, int S, const unsigned* ncols,
                              const int* p_conn, int cs, const int* p_elem, int mirror, const int* ids, const int* first_pos, int ntab,
                              uint2* lanes_tab, int4* hdr, int* mismatch, int max_row = 128);
// mirror != 0 (k_hex8_rows, hex8_rows.hip): a block whose two nodes are both owned by the position keeps the lanes of the smaller node's
// owner only (x bit 29: the lane also stores the transpose to the twin), and y = offset in doubles | node << 13 | twin offset << 16 |
// twin node << 29 (row strides from the position record)

// element records of the affine elements (elem_aff[e] != 0) among [e_first, e_end) from the current vertex coordinates: once per assembly, on the same
// stream right before affine_rows_launch.  A singular element (det J == 0 exactly) of the active set (active == NULL: all) is
// reported through `status`.
hipError_t affine_records_launch(int op, hipStream_t stream, const double* verts, const int* conn, const unsigned char* elem_aff,
                                 const unsigned char* active, long long e_first, long long e_end, double* rec, DevStatus* status);

// Positions with identical lane records share one table: `lanes_full` [npos][256] as written by affine_rows_build, `ids` the
// table of every position, `first_pos` [ntab] a position that holds each table.  Writes the compact tables, puts the id into
// every header, and sets *mismatch if the records of a position differ from its table (callers fall back to ids = identity).
hipError_t affine_rows_compact(hipStream_t stream, const uint2* lanes_full, const int* ids, const int* first_pos, int npos, int ntab,
                               uint2* lanes_tab, int4* hdr, int* mismatch);

// op: FH_LAPLACE or FH_LINEAR_ELASTIC; depth (1 or 2): positions the loader wave's requests run ahead; ablate != 0 selects the instrumented
// instantiation (profiling only).  (Retired to scripts/attic/: the fused form whose seventh wavefront formed the element records, the chunked
// dealing of positions, a second store wave, and the ring form affine_ring.hip.)
hipError_t affine_rows_launch(int op, int depth, int grid, size_t lds_bytes, hipStream_t stream, const KArgs& a, const AffineRowTables& T, int ablate,
                              bool masked);

}  // namespace fenris_hip
