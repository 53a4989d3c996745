// Host-side input generators of the engine: quadrature rules, structured mesh generators, Hex8->Hex27
// conversion, Lame conversion and the reference-identical greedy colouring.  Pure C++ (no device code);
// everything here reproduces the *orderings* of the reference bit-exactly, because CSR index parity
// depends on them.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <unordered_map>
#include <string>
#include <vector>

#include "../../include/fenris_hip.h"
#include "host_inputs.hpp"

namespace fenris_hip {

// ---------------------------------------------------------------------------------------- quadrature
// Gauss-Legendre by Newton iteration on P_n, following fenris-quadrature/src/univariate.rs:13-118:
// three-term recurrence for (P_n, P_{n-1}), derivative n (x P_n - P_{n-1}) / (x^2 - 1), start value
// cos(pi (i + 3/4) / (n + 1/2)), stop when |dx| <= 1e-15, weights 2 / ((1 - x^2) P_n'(x)^2); the first
// ceil(n/2) roots are the positive ones in descending order, the rest are mirrored.
static inline void legendre_pair(unsigned n, double x, double& pn, double& dpn) {
    double cur = 1.0, prev = 0.0;
    for (unsigned m = 1; m <= n; ++m) {
        const double mm = static_cast<double>(m);
        const double prev2 = prev;
        prev = cur;
        cur = ((2.0 * mm - 1.0) * x * prev - (mm - 1.0) * prev2) / mm;
    }
    pn = cur;
    dpn = static_cast<double>(n) * (x * cur - prev) / (x * x - 1.0);
}

bool gauss_rule(unsigned n, std::vector<double>& w, std::vector<double>& x) {
    if (n == 0) return false;
    w.assign(n, 0.0);
    x.assign(n, 0.0);
    const unsigned half = (n + 1) / 2;
    for (unsigned i = 0; i < half; ++i) {
        double xi = std::cos(M_PI * (static_cast<double>(i) + 0.75) / (static_cast<double>(n) + 0.5));
        double p, dp;
        legendre_pair(n, xi, p, dp);
        while (true) {
            const double step = -p / dp;
            xi += step;
            legendre_pair(n, xi, p, dp);
            if (std::fabs(step) <= 1e-15) break;
        }
        x[i] = xi;
        w[i] = 2.0 / ((1.0 - xi * xi) * dp * dp);
    }
    for (unsigned i = half; i < n; ++i) {
        x[i] = -x[n - 1 - i];
        w[i] = w[n - 1 - i];
    }
    return true;
}

// Tensor rules: first coordinate outermost, last innermost, w = wx*wy(*wz)  (tensor.rs:13-55)
bool tensor_rule(unsigned dim, unsigned n, double* w_out, double* p_out) {
    std::vector<double> w, x;
    if (!gauss_rule(n, w, x)) return false;
    size_t k = 0;
    if (dim == 2) {
        for (unsigned a = 0; a < n; ++a)
            for (unsigned b = 0; b < n; ++b, ++k) {
                w_out[k] = w[a] * w[b];
                p_out[2 * k] = x[a];
                p_out[2 * k + 1] = x[b];
            }
    } else {
        for (unsigned a = 0; a < n; ++a)
            for (unsigned b = 0; b < n; ++b)
                for (unsigned c = 0; c < n; ++c, ++k) {
                    w_out[k] = w[a] * w[b] * w[c];
                    p_out[3 * k] = x[a];
                    p_out[3 * k + 1] = x[b];
                    p_out[3 * k + 2] = x[c];
                }
    }
    return true;
}

// Witherden-Vincent tables as decimal strings, converted with strtod like Rust's str::parse::<f64>
// (polyquad-parse/src/lib.rs:48-51).  Row = coordinates..., weight.  Tabulated: tet 1, 2, 3, 5, 6; tri 1, 2, 4, 5, 6.
struct TableRule { unsigned strength, npts; const char* const* rows; };
#include "polyquad_tables.inc"
static const TableRule TET_RULES[] = {{FH_PQ_TET_TABLES[0].strength, FH_PQ_TET_TABLES[0].npts, FH_PQ_TET_TABLES[0].rows},
                                      {FH_PQ_TET_TABLES[1].strength, FH_PQ_TET_TABLES[1].npts, FH_PQ_TET_TABLES[1].rows},
                                      {FH_PQ_TET_TABLES[2].strength, FH_PQ_TET_TABLES[2].npts, FH_PQ_TET_TABLES[2].rows},
                                      {FH_PQ_TET_TABLES[3].strength, FH_PQ_TET_TABLES[3].npts, FH_PQ_TET_TABLES[3].rows},
                                      {FH_PQ_TET_TABLES[4].strength, FH_PQ_TET_TABLES[4].npts, FH_PQ_TET_TABLES[4].rows}};
static const TableRule TRI_RULES[] = {{FH_PQ_TRI_TABLES[0].strength, FH_PQ_TRI_TABLES[0].npts, FH_PQ_TRI_TABLES[0].rows},
                                      {FH_PQ_TRI_TABLES[1].strength, FH_PQ_TRI_TABLES[1].npts, FH_PQ_TRI_TABLES[1].rows},
                                      {FH_PQ_TRI_TABLES[2].strength, FH_PQ_TRI_TABLES[2].npts, FH_PQ_TRI_TABLES[2].rows},
                                      {FH_PQ_TRI_TABLES[3].strength, FH_PQ_TRI_TABLES[3].npts, FH_PQ_TRI_TABLES[3].rows},
                                      {FH_PQ_TRI_TABLES[4].strength, FH_PQ_TRI_TABLES[4].npts, FH_PQ_TRI_TABLES[4].rows}};

// select_minimum (fenris-quadrature/build.rs:172-194): smallest tabulated strength >= requested
template <size_t K>
static int table_rule(const TableRule (&rules)[K], unsigned dim, unsigned strength, double* w, double* p,
                      uint32_t* npts) {
    for (const TableRule& r : rules) {
        if (r.strength < strength) continue;
        for (unsigned i = 0; i < r.npts; ++i) {
            for (unsigned c = 0; c < dim; ++c) p[dim * i + c] = std::strtod(r.rows[(dim + 1) * i + c], nullptr);
            w[i] = std::strtod(r.rows[(dim + 1) * i + dim], nullptr);
        }
        *npts = r.npts;
        return FH_OK;
    }
    return FH_UNSUPPORTED;
}

// ---------------------------------------------------------------------------------------- meshes
// Structured generators; vertex and cell orders follow src/mesh/procedural.rs (A.6 of the survey).
void quad_mesh_sizes(uint64_t ux, uint64_t uy, uint64_t cpu, uint64_t& nv, uint64_t& nc) {
    if (!ux || !uy || !cpu) { nv = nc = 0; return; }
    nv = (ux * cpu + 1) * (uy * cpu + 1);
    nc = ux * cpu * uy * cpu;
}

void quad_mesh_fill(double unit, uint64_t ux, uint64_t uy, uint64_t cpu, const double tl[2], double* v,
                    uint64_t* c) {
    const double h = unit / static_cast<double>(cpu);
    const uint64_t nx = ux * cpu, ny = uy * cpu, stride = nx + 1;
    for (uint64_t j = 0; j <= ny; ++j)
        for (uint64_t i = 0; i <= nx; ++i) {
            double* out = v + 2 * (stride * j + i);
            out[0] = tl[0] + static_cast<double>(i) * h;
            out[1] = tl[1] + (-static_cast<double>(j)) * h;
        }
    for (uint64_t j = 0; j < ny; ++j)
        for (uint64_t i = 0; i < nx; ++i) {
            uint64_t* q = c + 4 * (nx * j + i);
            q[0] = stride * (j + 1) + i;
            q[1] = stride * (j + 1) + i + 1;
            q[2] = stride * j + i + 1;
            q[3] = stride * j + i;
        }
}

void hex_mesh_sizes(uint64_t ux, uint64_t uy, uint64_t uz, uint64_t cpu, uint64_t& nv, uint64_t& nc) {
    if (!ux || !uy || !cpu) { nv = nc = 0; return; }  // procedural.rs:236 (units_z is not checked there)
    nv = (ux * cpu + 1) * (uy * cpu + 1) * (uz * cpu + 1);
    nc = ux * cpu * uy * cpu * uz * cpu;
}

void hex_mesh_fill(double unit, uint64_t ux, uint64_t uy, uint64_t uz, uint64_t cpu, double* v, uint64_t* c) {
    const double h = unit / static_cast<double>(cpu);
    const uint64_t nx = ux * cpu, ny = uy * cpu, nz = uz * cpu;
    const uint64_t sx = nx + 1, sxy = (nx + 1) * (ny + 1);
    if (v) {
        double* out = v;
        for (uint64_t k = 0; k <= nz; ++k)
            for (uint64_t j = 0; j <= ny; ++j)
                for (uint64_t i = 0; i <= nx; ++i, out += 3) {
                    out[0] = static_cast<double>(i) * h;
                    out[1] = static_cast<double>(j) * h;
                    out[2] = static_cast<double>(k) * h;
                }
    }
    if (c) {
        // local corner (di,dj,dk) per Hex8 node: (0,0,0),(1,0,0),(1,1,0),(0,1,0),(0,0,1),(1,0,1),(1,1,1),(0,1,1)
        static const uint64_t CORNER[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0},
                                              {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};
        uint64_t* out = c;
        for (uint64_t k = 0; k < nz; ++k)
            for (uint64_t j = 0; j < ny; ++j)
                for (uint64_t i = 0; i < nx; ++i, out += 8)
                    for (int a = 0; a < 8; ++a)
                        out[a] = sxy * (k + CORNER[a][2]) + sx * (j + CORNER[a][1]) + (i + CORNER[a][0]);
    }
}

// BCC tetrahedral mesh (procedural.rs:286-403): lattice vertices, then cell centres; for every cell and
// axis: 4 tets around the centre-centre edge towards the +axis neighbour, and a 2-tet pyramid on each
// boundary face with the diagonal alternating by (i+j+k) % 2.
void tet_mesh_sizes(uint64_t ux, uint64_t uy, uint64_t uz, uint64_t cpu, uint64_t& nv, uint64_t& nc) {
    if (!ux || !uy || !uz || !cpu) { nv = nc = 0; return; }
    const uint64_t cx = ux * cpu, cy = uy * cpu, cz = uz * cpu;
    nv = (cx + 1) * (cy + 1) * (cz + 1) + cx * cy * cz;
    nc = 4 * ((cx - 1) * cy * cz + cx * (cy - 1) * cz + cx * cy * (cz - 1)) + 4 * (cy * cz + cx * cz + cx * cy);
}

void tet_mesh_fill(double unit, uint64_t ux, uint64_t uy, uint64_t uz, uint64_t cpu, double* v, uint64_t* c) {
    const double h = unit / static_cast<double>(cpu);
    const uint64_t cells[3] = {ux * cpu, uy * cpu, uz * cpu};
    const uint64_t vx = cells[0] + 1, vy = cells[1] + 1, vz = cells[2] + 1;
    const uint64_t centre0 = vx * vy * vz;
    double* out = v;
    for (uint64_t k = 0; k < vz; ++k)
        for (uint64_t j = 0; j < vy; ++j)
            for (uint64_t i = 0; i < vx; ++i, out += 3) {
                out[0] = h * static_cast<double>(i);
                out[1] = h * static_cast<double>(j);
                out[2] = h * static_cast<double>(k);
            }
    for (uint64_t k = 0; k < cells[2]; ++k)
        for (uint64_t j = 0; j < cells[1]; ++j)
            for (uint64_t i = 0; i < cells[0]; ++i, out += 3) {
                out[0] = h * (0.5 + static_cast<double>(i));
                out[1] = h * (0.5 + static_cast<double>(j));
                out[2] = h * (0.5 + static_cast<double>(k));
            }
    auto vid = [&](const std::array<int64_t, 3>& p) {
        return (vx * vy) * static_cast<uint64_t>(p[2]) + vx * static_cast<uint64_t>(p[1]) + static_cast<uint64_t>(p[0]);
    };
    auto cid = [&](uint64_t i, uint64_t j, uint64_t k) { return (cells[0] * cells[1]) * k + cells[0] * j + i + centre0; };
    // corners of the face shared with the +axis neighbour, as offsets from (i,j,k)
    static const int64_t FACE[3][4][3] = {{{1, 0, 1}, {1, 1, 1}, {1, 1, 0}, {1, 0, 0}},
                                          {{0, 1, 0}, {1, 1, 0}, {1, 1, 1}, {0, 1, 1}},
                                          {{0, 1, 1}, {1, 1, 1}, {1, 0, 1}, {0, 0, 1}}};
    uint64_t* t = c;
    auto emit = [&](uint64_t a, uint64_t b, uint64_t cc, uint64_t d) { t[0] = a; t[1] = b; t[2] = cc; t[3] = d; t += 4; };
    for (uint64_t k = 0; k < cells[2]; ++k)
        for (uint64_t j = 0; j < cells[1]; ++j)
            for (uint64_t i = 0; i < cells[0]; ++i) {
                const uint64_t ijk[3] = {i, j, k};
                for (int axis = 0; axis < 3; ++axis) {
                    std::array<std::array<int64_t, 3>, 4> f;
                    for (int q = 0; q < 4; ++q)
                        for (int r = 0; r < 3; ++r) f[q][r] = FACE[axis][q][r] + static_cast<int64_t>(ijk[r]);
                    if (ijk[axis] + 1 < cells[axis]) {
                        const uint64_t c1 = cid(i, j, k);
                        const uint64_t c2 = cid(i + (axis == 0), j + (axis == 1), k + (axis == 2));
                        for (int q = 0; q < 4; ++q) emit(c1, c2, vid(f[(q + 1) & 3]), vid(f[q]));
                    }
                    for (int positive = 0; positive < 2; ++positive) {
                        if (positive == 0 && ijk[axis] != 0) continue;
                        if (positive == 1 && ijk[axis] + 1 != cells[axis]) continue;
                        auto g = f;
                        if (!positive) {
                            std::reverse(g.begin(), g.end());
                            for (auto& p : g) p[axis] -= 1;
                        }
                        const uint64_t a = vid(g[0]), b = vid(g[1]), cc = vid(g[2]), d = vid(g[3]);
                        const uint64_t ctr = cid(i, j, k);
                        if ((i + j + k) % 2 == 0) { emit(a, b, cc, ctr); emit(a, cc, d, ctr); }
                        else { emit(a, b, d, ctr); emit(b, cc, d, ctr); }
                    }
                }
            }
}

// Hex8 -> Hex27 (mesh_convert.rs:85-166 + 227-330): per element 8 corners, 12 edge midpoints
// (lerp 0.5), 6 face points and the centre through the trilinear map; global numbering = first
// occurrence while sweeping elements in order, local nodes 0..26 in order, keyed by the sorted set of
// parent vertices.
static inline double trilinear_shape(int node, const double xi[3]) {
    static const double SGN[8][3] = {{-1, -1, -1}, {1, -1, -1}, {1, 1, -1}, {-1, 1, -1},
                                     {-1, -1, 1}, {1, -1, 1}, {1, 1, 1}, {-1, 1, 1}};
    return ((1.0 + SGN[node][0] * xi[0]) / 2.0) * ((1.0 + SGN[node][1] * xi[1]) / 2.0) * ((1.0 + SGN[node][2] * xi[2]) / 2.0);
}

struct ParentKey {
    std::array<uint64_t, 8> p;
    bool operator==(const ParentKey& o) const { return p == o.p; }
};
struct ParentKeyHash {
    size_t operator()(const ParentKey& k) const {
        uint64_t h = 0x9e3779b97f4a7c15ull;
        for (uint64_t x : k.p) { h ^= x + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2); }
        return static_cast<size_t>(h);
    }
};

void hex8_to_hex27(const double* verts, const uint64_t* hex8, uint64_t ncells, double* out_v, uint64_t* out_nv,
                   uint64_t* out_c) {
    static const int EDGE[12][2] = {{0, 1}, {0, 3}, {0, 4}, {1, 2}, {1, 5}, {2, 3}, {2, 6}, {3, 7}, {4, 5}, {4, 7}, {5, 6}, {6, 7}};
    static const int FACE[6][4] = {{0, 1, 2, 3}, {0, 1, 4, 5}, {0, 3, 4, 7}, {1, 2, 5, 6}, {2, 3, 6, 7}, {4, 5, 6, 7}};
    static const double FACE_XI[7][3] = {{0, 0, -1}, {0, -1, 0}, {-1, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {0, 0, 0}};
    const uint64_t NONE = ~0ull;
    std::unordered_map<ParentKey, uint64_t, ParentKeyHash> label;
    label.reserve(static_cast<size_t>(ncells) * 10);
    uint64_t next = 0;
    for (uint64_t e = 0; e < ncells; ++e) {
        const uint64_t* g = hex8 + 8 * e;
        double X[8][3];
        for (int a = 0; a < 8; ++a)
            for (int r = 0; r < 3; ++r) X[a][r] = verts[3 * g[a] + r];
        double pos[27][3];
        ParentKey key[27];
        for (auto& k : key) k.p.fill(NONE);
        for (int a = 0; a < 8; ++a) {
            std::memcpy(pos[a], X[a], sizeof(double) * 3);
            key[a].p[0] = g[a];
        }
        for (int m = 0; m < 12; ++m) {
            const int b = EDGE[m][0], en = EDGE[m][1];
            // nalgebra lerp: res = t * rhs + (1 - t) * self
            for (int r = 0; r < 3; ++r) pos[8 + m][r] = 0.5 * X[en][r] + (1.0 - 0.5) * X[b][r];
            key[8 + m].p[0] = g[b];
            key[8 + m].p[1] = g[en];
        }
        for (int f = 0; f < 7; ++f) {
            double N[8];
            for (int a = 0; a < 8; ++a) N[a] = trilinear_shape(a, FACE_XI[f]);
            for (int r = 0; r < 3; ++r) {
                double s = X[0][r] * N[0];
                for (int a = 1; a < 8; ++a) s = s + X[a][r] * N[a];
                pos[20 + f][r] = s;
            }
            if (f < 6) for (int q = 0; q < 4; ++q) key[20 + f].p[q] = g[FACE[f][q]];
            else for (int q = 0; q < 8; ++q) key[26].p[q] = g[q];
        }
        for (int a = 0; a < 27; ++a) {
            std::sort(key[a].p.begin(), key[a].p.end());  // NONE sorts last: padding stays at the end
            auto it = label.find(key[a]);
            if (it == label.end()) {
                label.emplace(key[a], next);
                std::memcpy(out_v + 3 * next, pos[a], sizeof(double) * 3);
                out_c[27 * e + a] = next++;
            } else {
                out_c[27 * e + a] = it->second;
            }
        }
    }
    *out_nv = next;
}

// ---------------------------------------------------------------------------------------- colouring
// sequential_greedy_coloring (fenris-paradis/src/coloring.rs:6-70): repeated passes over the still
// uncoloured elements in ascending order; an element joins the current colour iff none of its nodes
// already carries the current colour stamp; stamps are updated immediately.
void greedy_coloring(uint64_t E, const uint64_t* offs, const uint64_t* nodes, std::vector<uint64_t>& color_offsets,
                     std::vector<uint64_t>& labels) {
    uint64_t max_node = 0;
    for (uint64_t i = 0; i < offs[E]; ++i) max_node = std::max(max_node, nodes[i]);
    std::vector<int32_t> stamp(static_cast<size_t>(max_node) + 1, -1);
    std::vector<uint64_t> todo(E), later;
    for (uint64_t e = 0; e < E; ++e) todo[e] = e;
    labels.clear();
    labels.reserve(E);
    color_offsets.assign(1, 0);
    for (int32_t color = 0; !todo.empty(); ++color) {
        later.clear();
        for (uint64_t e : todo) {
            bool blocked = false;
            for (uint64_t k = offs[e]; k < offs[e + 1] && !blocked; ++k) blocked = (stamp[nodes[k]] == color);
            if (blocked) { later.push_back(e); continue; }
            for (uint64_t k = offs[e]; k < offs[e + 1]; ++k) stamp[nodes[k]] = color;
            labels.push_back(e);
        }
        color_offsets.push_back(labels.size());
        todo.swap(later);
    }
}

}  // namespace fenris_hip

// ------------------------------------------------------------------------------------------- C ABI
using namespace fenris_hip;

extern "C" {

int fh_gauss(uint32_t n, double* w, double* x) {
    std::vector<double> ww, xx;
    if (!w || !x || !gauss_rule(n, ww, xx)) return FH_BAD_ARGUMENT;
    std::copy(ww.begin(), ww.end(), w);
    std::copy(xx.begin(), xx.end(), x);
    return FH_OK;
}
int fh_quadrilateral_gauss(uint32_t n, double* w, double* p) {
    return (w && p && tensor_rule(2, n, w, p)) ? FH_OK : FH_BAD_ARGUMENT;
}
int fh_hexahedron_gauss(uint32_t n, double* w, double* p) {
    return (w && p && tensor_rule(3, n, w, p)) ? FH_OK : FH_BAD_ARGUMENT;
}
int fh_tetrahedron_rule(uint32_t strength, double* w, double* p, uint32_t* np) {
    if (!w || !p || !np) return FH_BAD_ARGUMENT;
    return table_rule(TET_RULES, 3, strength, w, p, np);
}
int fh_triangle_rule(uint32_t strength, double* w, double* p, uint32_t* np) {
    if (!w || !p || !np) return FH_BAD_ARGUMENT;
    return table_rule(TRI_RULES, 2, strength, w, p, np);
}

int fh_quad_mesh_2d(double unit, uint64_t ux, uint64_t uy, uint64_t cpu, const double tl[2], double* v, uint64_t* c,
                    uint64_t* nv, uint64_t* nc) {
    uint64_t a, b;
    quad_mesh_sizes(ux, uy, cpu, a, b);
    if (nv) *nv = a;
    if (nc) *nc = b;
    if (v && c && a) {
        if (!tl) return FH_BAD_ARGUMENT;
        quad_mesh_fill(unit, ux, uy, cpu, tl, v, c);
    }
    return FH_OK;
}
int fh_hex_mesh(double unit, uint64_t ux, uint64_t uy, uint64_t uz, uint64_t cpu, double* v, uint64_t* c, uint64_t* nv,
                uint64_t* nc) {
    uint64_t a, b;
    hex_mesh_sizes(ux, uy, uz, cpu, a, b);
    if (nv) *nv = a;
    if (nc) *nc = b;
    if ((v || c) && a) hex_mesh_fill(unit, ux, uy, uz, cpu, v, c);
    return FH_OK;
}
int fh_tet_mesh(double unit, uint64_t ux, uint64_t uy, uint64_t uz, uint64_t cpu, double* v, uint64_t* c, uint64_t* nv,
                uint64_t* nc) {
    uint64_t a, b;
    tet_mesh_sizes(ux, uy, uz, cpu, a, b);
    if (nv) *nv = a;
    if (nc) *nc = b;
    if (v && c && a) tet_mesh_fill(unit, ux, uy, uz, cpu, v, c);
    return FH_OK;
}
int fh_hex8_to_hex27(const double* v, uint64_t nv, const uint64_t* hex8, uint64_t ncells, double* out_v, uint64_t* out_nv,
                     uint64_t* out_c) {
    if (!v || !hex8 || !out_v || !out_nv || !out_c) return FH_BAD_ARGUMENT;
    for (uint64_t i = 0; i < 8 * ncells; ++i)
        if (hex8[i] >= nv) return FH_BAD_ARGUMENT;
    hex8_to_hex27(v, hex8, ncells, out_v, out_nv, out_c);
    return FH_OK;
}
// Tet20Mesh::from(&tet4_mesh), src/mesh_convert.rs:658-775
int fh_tet4_to_tet20(const double* v, uint64_t nv, const uint64_t* tet4, uint64_t ncells, double* out_v, uint64_t* out_nv,
                     uint64_t* out_c) {
    if (!v || !tet4 || !out_v || !out_nv || !out_c) return FH_BAD_ARGUMENT;
    for (uint64_t i = 0; i < 4 * ncells; ++i)
        if (tet4[i] >= nv) return FH_BAD_ARGUMENT;
    using Key = std::array<uint64_t, 4>;
    static const int ED[6][2] = {{0, 1}, {0, 2}, {0, 3}, {1, 2}, {1, 3}, {2, 3}};
    static const int FA[4][3] = {{0, 1, 2}, {0, 1, 3}, {0, 2, 3}, {1, 2, 3}};
    auto element_keys = [&](const uint64_t* g, Key (&k)[20]) {
        for (int a = 0; a < 4; ++a) k[a] = Key{g[a], 0, 0, 0};
        for (int m = 0; m < 6; ++m)
            for (uint64_t local = 0; local < 2; ++local) {
                uint64_t s0 = g[ED[m][0]], e0 = g[ED[m][1]], l = local;
                if (s0 > e0) { std::swap(s0, e0); l = (l + 1) % 2; }  // normalized_edge :685-692
                k[4 + 2 * m + (int)local] = Key{s0, e0, l, 1};
            }
        for (int f = 0; f < 4; ++f) {
            std::array<uint64_t, 3> t{g[FA[f][0]], g[FA[f][1]], g[FA[f][2]]};
            std::sort(t.begin(), t.end());
            k[16 + f] = Key{t[0], t[1], t[2], 2};
        }
    };
    std::vector<Key> all;
    all.reserve((size_t)ncells * 20);
    Key k[20];
    for (uint64_t e = 0; e < ncells; ++e) {
        element_keys(tet4 + 4 * e, k);
        all.insert(all.end(), k, k + 20);
    }
    std::sort(all.begin(), all.end());
    all.erase(std::unique(all.begin(), all.end()), all.end());
    for (uint64_t e = 0; e < ncells; ++e) {
        element_keys(tet4 + 4 * e, k);
        for (int a = 0; a < 20; ++a) out_c[20 * e + a] = (uint64_t)(std::lower_bound(all.begin(), all.end(), k[a]) - all.begin());
    }
    for (size_t i = 0; i < all.size(); ++i) {
        const Key& q = all[i];
        for (int r = 0; r < 3; ++r) {
            double x;
            if (q[3] == 0) {
                x = v[3 * q[0] + r];
            } else if (q[3] == 1) {  // start + (end - start) * ((local + 1) / 3)
                const double st = v[3 * q[0] + r], en = v[3 * q[1] + r];
                const double alpha = (double)(q[2] + 1) / 3.0;
                x = st + (en - st) * alpha;
            } else {                 // (a + b + c) / 3
                x = ((v[3 * q[0] + r] + v[3 * q[1] + r]) + v[3 * q[2] + r]) / 3.0;
            }
            out_v[3 * i + r] = x;
        }
    }
    *out_nv = all.size();
    return FH_OK;
}

// p-refinement of linear meshes to their quadratic counterparts (src/mesh_convert.rs)
int fh_refine_to_quadratic(int from_kind, const double* v, uint64_t nv, const uint64_t* conn, uint64_t ncells, double* out_v,
                           uint64_t* out_nv, uint64_t* out_c) {
    if (!v || !conn || !out_v || !out_nv || !out_c) return FH_BAD_ARGUMENT;
    const int n0 = (from_kind == FH_TET4) ? 4 : (from_kind == FH_TRI3) ? 3 : (from_kind == FH_QUAD4) ? 4 : (from_kind == FH_HEX8) ? 8 : 0;
    if (!n0) return FH_BAD_ARGUMENT;
    for (uint64_t i = 0; i < (uint64_t)n0 * ncells; ++i)
        if (conn[i] >= nv) return FH_BAD_ARGUMENT;
    if (from_kind == FH_TET4 || from_kind == FH_HEX8) {
        // Tet10Mesh::from(&tet4) / Hex20Mesh::from(&hex8): RefineFrom (mesh_convert.rs:42-83, 168-217) through the generic
        // relabelling of :227-330 -- every node (vertex nodes first, then the edge nodes) is identified by its sorted
        // parent vertices and labelled in order of first occurrence, so the old vertex indices are NOT kept
        static const int EDGE_T[6][2] = {{0, 1}, {1, 2}, {0, 2}, {0, 3}, {2, 3}, {1, 3}};
        static const int EDGE_H[12][2] = {{0, 1}, {0, 3}, {0, 4}, {1, 2}, {1, 5}, {2, 3}, {2, 6}, {3, 7}, {4, 5}, {4, 7}, {5, 6}, {6, 7}};
        const bool hex = (from_kind == FH_HEX8);
        const int nv0 = hex ? 8 : 4, ne = hex ? 12 : 6, n1 = nv0 + ne;
        const int (*EDGE)[2] = hex ? EDGE_H : EDGE_T;
        const uint64_t NONE = ~0ull;
        std::unordered_map<ParentKey, uint64_t, ParentKeyHash> label;
        label.reserve(static_cast<size_t>(ncells) * 4);
        uint64_t next = 0;
        for (uint64_t e = 0; e < ncells; ++e) {
            const uint64_t* g = conn + (uint64_t)nv0 * e;
            double pos[20][3];
            ParentKey key[20];
            for (auto& k : key) k.p.fill(NONE);
            for (int a = 0; a < nv0; ++a) {
                for (int r = 0; r < 3; ++r) pos[a][r] = v[3 * g[a] + r];
                key[a].p[0] = g[a];
            }
            for (int m = 0; m < ne; ++m) {
                const int b = EDGE[m][0], en = EDGE[m][1];
                // nalgebra lerp: self * (1 - t) + rhs * t
                for (int r = 0; r < 3; ++r) pos[nv0 + m][r] = v[3 * g[b] + r] * (1.0 - 0.5) + v[3 * g[en] + r] * 0.5;
                key[nv0 + m].p[0] = g[b];
                key[nv0 + m].p[1] = g[en];
            }
            for (int a = 0; a < n1; ++a) {
                std::sort(key[a].p.begin(), key[a].p.end());
                auto it = label.find(key[a]);
                if (it == label.end()) {
                    label.emplace(key[a], next);
                    std::memcpy(out_v + 3 * next, pos[a], sizeof(double) * 3);
                    out_c[(uint64_t)n1 * e + a] = next++;
                } else {
                    out_c[(uint64_t)n1 * e + a] = it->second;
                }
            }
        }
        *out_nv = next;
        return FH_OK;
    }
    // Tri6 from Tri3 (mesh_convert.rs:332-383) and Quad9 from Quad4 (:385-442): the vertices are kept, the midpoint of
    // every edge (a, b) -- consecutive vertices of the element, cyclically -- is appended at its first occurrence
    // ((v_a + v_b) / 2), Quad9 then appends the image of the reference origin
    const int n1 = (from_kind == FH_TRI3) ? 6 : 9;
    std::memcpy(out_v, v, sizeof(double) * 2 * nv);
    uint64_t next = nv;
    std::map<std::pair<uint64_t, uint64_t>, uint64_t> edge_index;
    for (uint64_t e = 0; e < ncells; ++e) {
        const uint64_t* g = conn + (uint64_t)n0 * e;
        uint64_t* o = out_c + (uint64_t)n1 * e;
        for (int a = 0; a < n0; ++a) o[a] = g[a];
        for (int m = 0; m < n0; ++m) {
            const uint64_t a = g[m], b = g[(m + 1) % n0];
            const auto key = std::make_pair(std::min(a, b), std::max(a, b));
            auto it = edge_index.find(key);
            if (it == edge_index.end()) {
                for (int r = 0; r < 2; ++r) out_v[2 * next + r] = (out_v[2 * a + r] + out_v[2 * b + r]) / 2.0;
                it = edge_index.emplace(key, next++).first;
            }
            o[n0 + m] = it->second;
        }
        if (from_kind == FH_QUAD4) {
            // map_reference_coords(origin) = X N^T with N_k = (1 + 0)(1 + 0) / 4 (quadrilateral.rs:117-122), k ascending
            for (int r = 0; r < 2; ++r) {
                double sacc = v[2 * g[0] + r] * 0.25;
                for (int k = 1; k < 4; ++k) sacc = sacc + v[2 * g[k] + r] * 0.25;
                out_v[2 * next + r] = sacc;
            }
            o[8] = next++;
        }
    }
    *out_nv = next;
    return FH_OK;
}

// cuthill_mckee (src/mesh/reorder.rs:171-233): breadth-first search from the first unvisited vertex of least degree,
// neighbours visited by ascending degree.  The reference sorts the neighbours with sort_unstable_by_key, which leaves
// the order of equal-degree neighbours unspecified; ties are resolved here by ascending vertex index (a stable sort
// of the ascending column list), which reproduces the reference's known answers (tests/unit_tests/reorder.rs).
int fh_cuthill_mckee(uint64_t n, const uint64_t* row_offsets, const uint64_t* col_indices, uint64_t* perm_out) {
    if (!row_offsets || !perm_out || (n && !col_indices && row_offsets[n])) return FH_BAD_ARGUMENT;
    std::vector<unsigned char> visited((size_t)n, 0);
    std::vector<uint64_t> queue;
    queue.reserve((size_t)n);
    std::vector<uint64_t> ws;
    auto degree = [&](uint64_t v) { return row_offsets[v + 1] - row_offsets[v]; };
    // unvisited vertices by ascending (degree, index): the restart rule "first unvisited vertex of least degree"
    // (reorder.rs:196-198) without the reference's O(N) rescan per component
    std::vector<uint64_t> by_degree((size_t)n);
    for (uint64_t i = 0; i < n; ++i) by_degree[(size_t)i] = i;
    std::stable_sort(by_degree.begin(), by_degree.end(), [&](uint64_t a, uint64_t b) { return degree(a) < degree(b); });
    size_t next_start = 0, head = 0;
    uint64_t count = 0;
    while (count < n) {
        while (visited[(size_t)by_degree[next_start]]) ++next_start;
        const uint64_t start = by_degree[next_start];
        visited[(size_t)start] = 1;
        queue.push_back(start);
        while (head < queue.size()) {
            const uint64_t v = queue[head++];
            ws.assign(col_indices + row_offsets[v], col_indices + row_offsets[v + 1]);
            for (uint64_t c : ws)
                if (c >= n) return FH_BAD_ARGUMENT;
            std::stable_sort(ws.begin(), ws.end(), [&](uint64_t a, uint64_t b) { return degree(a) < degree(b); });
            perm_out[count++] = v;
            for (uint64_t a : ws)
                if (!visited[(size_t)a]) { visited[(size_t)a] = 1; queue.push_back(a); }
        }
    }
    return FH_OK;
}

// reorder_mesh_par (src/mesh/reorder.rs:54-95): reverse Cuthill-McKee on the vertex graph (the solution-dim-1
// pattern of the mesh), then the elements stably sorted by their smallest NEW vertex index.
int fh_reorder_mesh(uint64_t num_vertices, uint64_t nodes_per_element, const uint64_t* connectivity, uint64_t num_elements,
                    uint64_t* vertex_perm, uint64_t* connectivity_perm) {
    if (!connectivity || !vertex_perm || !connectivity_perm || nodes_per_element == 0) return FH_BAD_ARGUMENT;
    const uint64_t N = num_vertices, n = nodes_per_element, E = num_elements;
    for (uint64_t i = 0; i < n * E; ++i)
        if (connectivity[i] >= N) return FH_BAD_ARGUMENT;
    // vertex graph = assemble_pattern of the mesh (global.rs:65-120 with solution dim 1): node -> sorted unique neighbours
    std::vector<uint64_t> deg((size_t)N + 1, 0);
    for (uint64_t i = 0; i < n * E; ++i) deg[(size_t)connectivity[i] + 1] += n;
    for (uint64_t i = 0; i < N; ++i) deg[(size_t)i + 1] += deg[(size_t)i];
    std::vector<uint64_t> raw((size_t)deg[(size_t)N]), fill(deg.begin(), deg.end() - 1);
    for (uint64_t e = 0; e < E; ++e)
        for (uint64_t a = 0; a < n; ++a) {
            const uint64_t i = connectivity[e * n + a];
            for (uint64_t b = 0; b < n; ++b) raw[(size_t)fill[(size_t)i]++] = connectivity[e * n + b];
        }
    std::vector<uint64_t> ro((size_t)N + 1, 0), ci;
    ci.reserve(raw.size() / 2);
    for (uint64_t i = 0; i < N; ++i) {
        auto b = raw.begin() + (std::ptrdiff_t)deg[(size_t)i], e2 = raw.begin() + (std::ptrdiff_t)deg[(size_t)i + 1];
        std::sort(b, e2);
        auto u = std::unique(b, e2);
        ci.insert(ci.end(), b, u);
        ro[(size_t)i + 1] = ci.size();
    }
    int rc = fh_cuthill_mckee(N, ro.data(), ci.data(), vertex_perm);
    if (rc) return rc;
    std::reverse(vertex_perm, vertex_perm + N);  // reverse_cuthill_mckee (reorder.rs:235-239)
    std::vector<uint64_t> inv((size_t)N);
    for (uint64_t t = 0; t < N; ++t) inv[(size_t)vertex_perm[t]] = t;  // old -> new index (Permutation::inverse)
    std::vector<uint64_t> key((size_t)E);
    for (uint64_t e = 0; e < E; ++e) {
        uint64_t m = UINT64_MAX;
        for (uint64_t a = 0; a < n; ++a) m = std::min(m, inv[(size_t)connectivity[e * n + a]]);
        key[(size_t)e] = m;
    }
    for (uint64_t e = 0; e < E; ++e) connectivity_perm[e] = e;
    std::stable_sort(connectivity_perm, connectivity_perm + E, [&](uint64_t a, uint64_t b) { return key[(size_t)a] < key[(size_t)b]; });
    return FH_OK;
}

// ---- Gmsh MSH 4.1 (ASCII) loader: load_msh_from_bytes, src/io/msh.rs:47-111 ---------------------------------------
// The reference parses with the mshio crate and then (a) concatenates the vertices of all node blocks in file order,
// refusing blocks whose tags are not consecutive, (b) concatenates the elements of every block whose (element type,
// entity dimension) matches the requested connectivity, node tags minus one, no reordering (msh.rs:232-274).
namespace {
thread_local std::string g_msh_error;
int msh_fail(const std::string& m) { g_msh_error = m; return FH_BAD_ARGUMENT; }
int msh_nodes_of_type(long t) {  // Gmsh element type -> node count (the types a 4.1 file commonly carries)
    switch (t) {
        case 1: return 2;  case 2: return 3;  case 3: return 4;  case 4: return 4;  case 5: return 8;  case 6: return 6;
        case 7: return 5;  case 8: return 3;  case 9: return 6;  case 10: return 9; case 11: return 10; case 12: return 27;
        case 13: return 18; case 14: return 14; case 15: return 1; case 16: return 8; case 17: return 20; case 18: return 15;
        case 19: return 13; case 20: return 9; case 21: return 10; case 26: return 4; case 29: return 20; case 92: return 64;
        default: return -1;
    }
}
struct MshCursor {
    const char* p;
    const char* end;
    bool token(std::string& out) {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p;
        if (p >= end) return false;
        const char* b = p;
        while (p < end && !(*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p;
        out.assign(b, p);
        return true;
    }
    bool integer(long long& v) {
        std::string t;
        if (!token(t)) return false;
        char* e = nullptr;
        v = std::strtoll(t.c_str(), &e, 10);
        return e && *e == 0;
    }
    bool real(double& v) {
        std::string t;
        if (!token(t)) return false;
        char* e = nullptr;
        v = std::strtod(t.c_str(), &e);
        return e && *e == 0;
    }
};
}  // namespace

const char* fh_msh_last_error(void) { return g_msh_error.c_str(); }

int fh_load_msh(const char* bytes, uint64_t len, int elem_kind, double* vertices, uint64_t* num_vertices, uint64_t* connectivity,
                uint64_t* num_elements) {
    if (!bytes || !num_vertices || !num_elements) return msh_fail("null argument");
    long want_type;
    int want_dim, gdim, nper;
    switch (elem_kind) {  // impl_msh_connectivity!, msh.rs:276-284
        case FH_TRI3: want_type = 2; want_dim = 2; gdim = 2; nper = 3; break;
        case FH_QUAD4: want_type = 3; want_dim = 2; gdim = 2; nper = 4; break;
        case FH_TET4: want_type = 4; want_dim = 3; gdim = 3; nper = 4; break;
        case FH_HEX8: want_type = 5; want_dim = 3; gdim = 3; nper = 8; break;
        case FH_HEX27: want_type = 12; want_dim = 3; gdim = 3; nper = 27; break;
        case FH_TET10: want_type = 11; want_dim = 3; gdim = 3; nper = 10; break;
        case FH_QUAD9: want_type = 10; want_dim = 2; gdim = 2; nper = 9; break;
        case FH_TRI6: want_type = 9; want_dim = 2; gdim = 2; nper = 6; break;
        default: return msh_fail("unsupported element kind");
    }
    MshCursor c{bytes, bytes + len};
    std::string tok;
    bool have_format = false, have_nodes = false, have_elements = false, matched = false;
    uint64_t nv = 0, ne = 0;
    long long claimed_nodes = 0;
    while (c.token(tok)) {
        if (tok == "$MeshFormat") {
            double ver;
            long long ftype, dsize;
            if (!c.real(ver) || !c.integer(ftype) || !c.integer(dsize)) return msh_fail("failed to parse msh file: bad $MeshFormat");
            if (ver < 4.1 || ver >= 4.2) return msh_fail("failed to parse msh file: only MSH 4.1 is supported");
            if (ftype != 0) return msh_fail("failed to parse msh file: binary MSH files are not supported by this loader");
            have_format = true;
        } else if (tok == "$Nodes") {
            long long nblocks, nn, mn, mx;
            if (!c.integer(nblocks) || !c.integer(nn) || !c.integer(mn) || !c.integer(mx)) return msh_fail("failed to parse msh file: bad $Nodes header");
            claimed_nodes = nn;
            for (long long b = 0; b < nblocks; ++b) {
                long long edim, etag, param, cnt;
                if (!c.integer(edim) || !c.integer(etag) || !c.integer(param) || !c.integer(cnt)) return msh_fail("failed to parse msh file: bad node block");
                long long first = 0;
                for (long long k = 0; k < cnt; ++k) {
                    long long tag;
                    if (!c.integer(tag)) return msh_fail("failed to parse msh file: bad node tag");
                    if (k == 0) first = tag;
                    // mshio keeps explicit tags only for non-consecutive blocks; the reference refuses those (msh.rs:117-122)
                    if (tag != first + k) return msh_fail("node block tags are not consecutive in msh file (sparse tags are not supported)");
                    // element nodes are addressed as tag - 1 (msh.rs:264): the k-th vertex read must carry tag k
                    if ((uint64_t)tag != nv + (uint64_t)k + 1) return msh_fail("node tags do not follow the file order (sparse tags are not supported)");
                }
                for (long long k = 0; k < cnt; ++k) {
                    double xyz[3], skip;
                    if (!c.real(xyz[0]) || !c.real(xyz[1]) || !c.real(xyz[2])) return msh_fail("failed to parse msh file: bad node coordinates");
                    for (long long q = 0; q < (param ? edim : 0); ++q)
                        if (!c.real(skip)) return msh_fail("failed to parse msh file: bad parametric coordinates");
                    if (vertices)
                        for (int d = 0; d < gdim; ++d) vertices[(nv + (uint64_t)k) * gdim + d] = xyz[d];  // point_from_msh_node :206-224
                }
                nv += (uint64_t)cnt;
            }
            have_nodes = true;
        } else if (tok == "$Elements") {
            long long nblocks, nel, mn, mx;
            if (!c.integer(nblocks) || !c.integer(nel) || !c.integer(mn) || !c.integer(mx)) return msh_fail("failed to parse msh file: bad $Elements header");
            for (long long b = 0; b < nblocks; ++b) {
                long long edim, etag, etype, cnt;
                if (!c.integer(edim) || !c.integer(etag) || !c.integer(etype) || !c.integer(cnt)) return msh_fail("failed to parse msh file: bad element block");
                const int nn = msh_nodes_of_type((long)etype);
                if (nn < 0) return msh_fail("failed to parse msh file: unknown element type");
                const bool take = (etype == want_type && edim == want_dim);  // element_block_matches_connectivity :176-187
                matched = matched || take;
                long long first = 0;
                for (long long k = 0; k < cnt; ++k) {
                    long long tag;
                    if (!c.integer(tag)) return msh_fail("failed to parse msh file: bad element tag");
                    if (k == 0) first = tag;
                    if (tag != first + k) return msh_fail("element block tags are not consecutive in msh file (sparse tags are not supported)");
                    for (int j = 0; j < nn; ++j) {
                        long long node;
                        if (!c.integer(node)) return msh_fail("failed to parse msh file: bad element nodes");
                        if (take && connectivity) connectivity[(ne + (uint64_t)k) * nper + j] = (uint64_t)(node - 1);
                    }
                }
                if (take) ne += (uint64_t)cnt;
            }
            have_elements = true;
        } else if (tok.size() > 1 && tok[0] == '$' && tok.compare(0, 4, "$End") != 0) {
            // any other section: skip to its end marker
            const std::string endm = "$End" + tok.substr(1);
            while (c.token(tok) && tok != endm) {}
        }
    }
    if (!have_format) return msh_fail("failed to parse msh file: no $MeshFormat section");
    if (!have_nodes) return msh_fail("MSH file does not contain nodes");
    if (!have_elements) return msh_fail("MSH file does not contain elements");
    if (!matched) return msh_fail("MSH file does not contain an element block of the requested type");
    if ((long long)nv != claimed_nodes) return msh_fail("fewer vertices were read than the msh file claims to contain");
    if (connectivity)
        for (uint64_t i = 0; i < ne * (uint64_t)nper; ++i)
            if (connectivity[i] >= nv) return msh_fail("element references a node that does not exist");
    *num_vertices = nv;
    *num_elements = ne;
    g_msh_error.clear();
    return FH_OK;
}

int fh_lame_from_young_poisson(double young, double poisson, double* mu, double* lambda) {
    if (!mu || !lambda) return FH_BAD_ARGUMENT;
    // fenris-solid/src/materials.rs:36-42
    const double m = 0.5 * young / (1.0 + poisson);
    *mu = m;
    *lambda = 2.0 * m * poisson / (1.0 - 2.0 * poisson);
    return FH_OK;
}

}  // extern "C"
