// General (non-affine) Hex8 form of the row-owner stiffness kernel, k_hex8_rows: interface of the translation unit hex8_rows.hip.
// Laplace / uniform LinearElastic on Hex8 with the eight-point rule (hexahedron_gauss(2)) -- what every hexahedral mesh that is not a box
// of parallelepipeds runs.  See hex8_rows.hip for the design; engine.hip builds the tables and launches.
#pragma once
#include <hip/hip_runtime.h>

#include "device_common.hpp"

namespace fenris_hip {

// position = block of up to seven consecutive nodes, in the sweep order of the pipelined kernel's tables (chains of blocks whose
// consecutive members share elements: the gradients of the shared elements stay staged in LDS)
struct Hex8RowTables {
    const int4* pos;      // [npos][4]    16 words per position: r0, nrow, flags | lane table << 8, new slots | occupied slots << 8,
                          //              32 slot bytes (the new slots first, then the retained), mask of the new slots, mask of the
                          //              occupied slots, 8 bytes: column blocks per row of the position's nodes
    const uint2* lanes;   // [ntab][256]  lane records (format of affine_rows.hip); positions with identical records share a table
    const int* conn;      // [npos][cs]   vertex index per (slot, local node)
    const int* elem;      // [npos][us]   element id per slot (error reporting)
    int us, cs, npos, acc_max;
};

constexpr int HEX8_ROWS_THREADS = 384;   // four row waves + one loader wave + one store wave
constexpr int HEX8_ROWS_US = 32;         // slots per position the kernel is laid out for

size_t hex8_rows_lds_bytes(int acc_max);

// position records from the pipelined kernel's records (p_rec: GatherHdr with k0 = number of new slots, slot list) and the headers the
// lane builder wrote (affine_rows_build + affine_rows_compact: {r0, nrow, flags | table << 8, U})
hipError_t hex8_rows_positions(hipStream_t stream, const int* p_rec, int rw, int us, int ms, const int4* hdr, int npos, int4* pos);

// Host side, once per pattern: rearranges the lanes of every table (`tables`: ntab x 256 records) so that the sixteen lanes the LDS serves
// together (ds_read_b128 lane groups) read operand vectors from different banks.  Only moves that leave every sum unchanged: whole
// DPP groups (aligned quads / pairs / single lanes) change places, and the two terms of a lane swap halves (the kernel adds its two
// term accumulators, a commutative sum).  Returns the modelled LDS cycles per position before / after (diagnostics).
void hex8_rows_tune_lanes(uint2* tables, int ntab, unsigned seed, double* cycles_before, double* cycles_after, long long total_proposals = 3000000ll);

// op: FH_LAPLACE or FH_LINEAR_ELASTIC; a.ggeom / a.qw: reference gradients [8][8][3] and weights [8] of the rule
constexpr int HEX8_ROWS_PRIO_SHIFT = 24;   // bits 24-29 of `ablate`: s_setprio level of the store wave, the loader wave, the row waves that carry phase B
hipError_t hex8_rows_launch(int op, int grid, size_t lds_bytes, hipStream_t stream, const KArgs& a, const Hex8RowTables& T, int ablate);

}  // namespace fenris_hip
