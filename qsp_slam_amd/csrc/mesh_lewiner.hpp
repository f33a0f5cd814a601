// mesh_lewiner.hpp -- Lewiner's marching cubes on the device: the algorithm the reference's mesh extraction runs
// (reconstruct/utils.py:131: skimage.measure.marching_cubes_lewiner(volume, level=0.0, spacing=[2/(n-1)]*3), then + (-1,-1,-1)).
// Lewiner, Lopes, Vieira, Tavares, "Efficient implementation of Marching Cubes' cases with topological guarantees" (JGT 2003), in
// the form scikit-image 0.18 ships it; same vertices, same faces, SAME ORDER as that implementation (tests/golden/mc_lewiner_*.npz
// hold its output; oracle/mc_lewiner_oracle.py is the CPU restatement the tests compare with).
//
// What fixes the output, and how it is reproduced without the serial sweep the CPU code does:
//   * axes (z, y, x) = the volume's (0, 1, 2); cells visited z-outermost: a cell's place in the sweep is the linear index of its
//     lowest corner;  corner positive <=> value > 0;  corners 0..3 = (x,y), (x+1,y), (x+1,y+1), (x,y+1) at z, 4..7 at z+1;
//   * per cell the case table gives (case, configuration); ambiguous cases are resolved by test_face / test_internal on the corner
//     values (in double); the chosen tiling is a row of edge ids 0..11 (12 = the extra vertex inside the cell);
//   * a vertex exists once; the CPU code creates it when a triangle first refers to it, so vertices are numbered by FIRST USE in the
//     sweep.  Every cell around a sign-changing edge uses that edge's vertex, so the first user is the cell with the smallest
//     coordinates among the (up to four) cells around the edge -- known from the coordinates alone.  A cell therefore knows which
//     entries of its own list create vertices and in which order: counts per cell, one exclusive scan, and every vertex has the
//     number the sweep would have given it;
//   * vertex position: sum_c w_c corner_c / sum_c w_c over the edge's two corners, w = 1 / (2.2e-16 + |value|), in double, stored
//     as float32 INDEX coordinates; the extra vertex the same over all eight corners, in corner order;
//   * faces: the tiling's triangles with their corners reversed (gradient_direction = 'descent').
// Three launches over the grid points around one scan: counts; vertices + the (edge -> vertex) map; faces.
#pragma once
#include "mc_lewiner_tables.hpp"

namespace qsp {
namespace lew {

constexpr double EPS = 2.220446049250313e-16;      // scikit-image's "FLT_EPSILON" is np.spacing(1.0)

// edge e of a cell: axis it runs along (0 = x, 1 = y, 2 = z), its lower corner's offset (dx, dy, dz), its two corners (Lewiner numbering)
__device__ const int8_t E_AXIS[12] = {0, 1, 0, 1, 0, 1, 0, 1, 2, 2, 2, 2};
__device__ const int8_t E_LO[12][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 0}, {0, 0, 1}, {1, 0, 1}, {0, 1, 1}, {0, 0, 1},
                                       {0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}};
__device__ const int8_t E_C1[12] = {0, 1, 2, 3, 4, 5, 6, 7, 0, 1, 2, 3};
__device__ const int8_t E_C2[12] = {1, 2, 3, 0, 5, 6, 7, 4, 4, 5, 6, 7};
__device__ const int8_t C_OFF[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};

__device__ inline bool test_face(int face, const double* v) {
#pragma clang fp contract(off)      // (products and differences rounded one by one, as the CPU code's are)
    const int f = face < 0 ? -face : face;
    double A, B, C, D;
    switch (f) {
        case 1: A = v[0], B = v[4], C = v[5], D = v[1]; break;
        case 2: A = v[1], B = v[5], C = v[6], D = v[2]; break;
        case 3: A = v[2], B = v[6], C = v[7], D = v[3]; break;
        case 4: A = v[3], B = v[7], C = v[4], D = v[0]; break;
        case 5: A = v[0], B = v[3], C = v[2], D = v[1]; break;
        default: A = v[4], B = v[7], C = v[6], D = v[5]; break;
    }
    const double acbd = A * C - B * D;
    if (acbd > -EPS && acbd < EPS) return face >= 0;
    return (double)face * A * acbd >= 0;
}

// corners (a, b) of the reference edge, then the three edges "parallel" to it as (from, to) pairs
__device__ const int8_t TI_EDGE[12][8] = {{0, 1, 3, 2, 7, 6, 4, 5}, {1, 2, 0, 3, 4, 7, 5, 6}, {2, 3, 1, 0, 5, 4, 6, 7}, {3, 0, 2, 1, 6, 5, 7, 4},
                                          {4, 5, 7, 6, 3, 2, 0, 1}, {5, 6, 4, 7, 0, 3, 1, 2}, {6, 7, 5, 4, 1, 0, 2, 3}, {7, 4, 6, 5, 2, 1, 3, 0},
                                          {0, 4, 3, 7, 2, 6, 1, 5}, {1, 5, 0, 4, 3, 7, 2, 6}, {2, 6, 1, 5, 0, 4, 3, 7}, {3, 7, 2, 6, 1, 5, 0, 4}};

__device__ inline bool test_internal(int cs, int config, int subconfig, int s, const double* v) {
#pragma clang fp contract(off)
    double At, Bt, Ct, Dt;
    if (cs == 4 || cs == 10) {
        const double a = (v[4] - v[0]) * (v[6] - v[2]) - (v[7] - v[3]) * (v[5] - v[1]);
        const double b = v[2] * (v[4] - v[0]) + v[0] * (v[6] - v[2]) - v[1] * (v[7] - v[3]) - v[3] * (v[5] - v[1]);
        const double t = -b / (2 * a);
        if (t < 0 || t > 1) return s > 0;
        At = v[0] + (v[4] - v[0]) * t;
        Bt = v[3] + (v[7] - v[3]) * t;
        Ct = v[2] + (v[6] - v[2]) * t;
        Dt = v[1] + (v[5] - v[1]) * t;
    } else {
        int edge;
        if (cs == 6) edge = TEST6[config * 3 + 2];
        else if (cs == 7) edge = TEST7[config * 5 + 4];
        else if (cs == 12) edge = TEST12[config * 4 + 3];
        else edge = TILING13_5_1[(config * 4 + subconfig) * 18];
        const int8_t* E = TI_EDGE[edge];
        const double t = v[E[0]] / (v[E[0]] - v[E[1]]);
        At = 0;
        Bt = v[E[2]] + (v[E[3]] - v[E[2]]) * t;
        Ct = v[E[4]] + (v[E[5]] - v[E[4]]) * t;
        Dt = v[E[6]] + (v[E[7]] - v[E[6]]) * t;
    }
    const int test = (At >= 0 ? 1 : 0) + (Bt >= 0 ? 2 : 0) + (Ct >= 0 ? 4 : 0) + (Dt >= 0 ? 8 : 0);
    switch (test) {
        case 0: case 1: case 2: case 3: case 4: case 6: case 8: case 9: case 12: return s > 0;
        case 7: case 11: case 13: case 14: case 15: return s < 0;
        // (Lewiner's C code returns s < 0 when the saddle condition of 5 / 10 fails; scikit-image's if / elif chain ends without a
        //  value there -- false whatever s is -- and that is what the reference runs: measured, oracle/gen_golden_mc.py)
        case 5: return (At * Ct - Bt * Dt < EPS) ? s > 0 : false;
        default: return (At * Ct - Bt * Dt >= EPS) ? s > 0 : false;
    }
}

// the tiling of one cell: pointer to its edge ids (three per triangle) and the number of triangles
__device__ inline int cell_tiling(int index, const double* v, const int8_t** row) {
    const int cs = CASES[2 * index], c = CASES[2 * index + 1];
#define QSP_LEW_ROW(table, width, r, n) do { *row = table + (size_t)(r) * (width); return (n); } while (0)
    switch (cs) {
        case 1: QSP_LEW_ROW(TILING1, 3, c, 1);
        case 2: QSP_LEW_ROW(TILING2, 6, c, 2);
        case 3:
            if (test_face(TEST3[c], v)) QSP_LEW_ROW(TILING3_2, 12, c, 4);
            QSP_LEW_ROW(TILING3_1, 6, c, 2);
        case 4:
            if (test_internal(cs, c, 0, TEST4[c], v)) QSP_LEW_ROW(TILING4_1, 6, c, 2);
            QSP_LEW_ROW(TILING4_2, 18, c, 6);
        case 5: QSP_LEW_ROW(TILING5, 9, c, 3);
        case 6:
            if (test_face(TEST6[3 * c], v)) QSP_LEW_ROW(TILING6_2, 15, c, 5);
            if (test_internal(cs, c, 0, TEST6[3 * c + 1], v)) QSP_LEW_ROW(TILING6_1_1, 9, c, 3);
            QSP_LEW_ROW(TILING6_1_2, 27, c, 9);
        case 7: {
            int sub = 0;
            if (test_face(TEST7[5 * c], v)) sub += 1;
            if (test_face(TEST7[5 * c + 1], v)) sub += 2;
            if (test_face(TEST7[5 * c + 2], v)) sub += 4;
            switch (sub) {
                case 0: QSP_LEW_ROW(TILING7_1, 9, c, 3);
                case 1: QSP_LEW_ROW(TILING7_2, 15, 3 * c + 0, 5);
                case 2: QSP_LEW_ROW(TILING7_2, 15, 3 * c + 1, 5);
                case 3: QSP_LEW_ROW(TILING7_3, 27, 3 * c + 0, 9);
                case 4: QSP_LEW_ROW(TILING7_2, 15, 3 * c + 2, 5);
                case 5: QSP_LEW_ROW(TILING7_3, 27, 3 * c + 1, 9);
                case 6: QSP_LEW_ROW(TILING7_3, 27, 3 * c + 2, 9);
                default:
                    if (test_internal(cs, c, 0, TEST7[5 * c + 3], v)) QSP_LEW_ROW(TILING7_4_2, 27, c, 9);
                    QSP_LEW_ROW(TILING7_4_1, 15, c, 5);
            }
        }
        case 8: QSP_LEW_ROW(TILING8, 6, c, 2);
        case 9: QSP_LEW_ROW(TILING9, 12, c, 4);
        case 10:
            if (test_face(TEST10[3 * c], v)) {
                if (test_face(TEST10[3 * c + 1], v)) QSP_LEW_ROW(TILING10_1_1_, 12, c, 4);
                QSP_LEW_ROW(TILING10_2, 24, c, 8);
            }
            if (test_face(TEST10[3 * c + 1], v)) QSP_LEW_ROW(TILING10_2_, 24, c, 8);
            if (test_internal(cs, c, 0, TEST10[3 * c + 2], v)) QSP_LEW_ROW(TILING10_1_1, 12, c, 4);
            QSP_LEW_ROW(TILING10_1_2, 24, c, 8);
        case 11: QSP_LEW_ROW(TILING11, 12, c, 4);
        case 12:
            if (test_face(TEST12[4 * c], v)) {
                if (test_face(TEST12[4 * c + 1], v)) QSP_LEW_ROW(TILING12_1_1_, 12, c, 4);
                QSP_LEW_ROW(TILING12_2, 24, c, 8);
            }
            if (test_face(TEST12[4 * c + 1], v)) QSP_LEW_ROW(TILING12_2_, 24, c, 8);
            if (test_internal(cs, c, 0, TEST12[4 * c + 2], v)) QSP_LEW_ROW(TILING12_1_1, 12, c, 4);
            QSP_LEW_ROW(TILING12_1_2, 24, c, 8);
        case 13: {
            int sub = 0;
            for (int k = 0; k < 6; ++k)
                if (test_face(TEST13[7 * c + k], v)) sub |= 1 << k;
            const int sc = SUBCONFIG13[sub];
            if (sc == 0) QSP_LEW_ROW(TILING13_1, 12, c, 4);
            if (sc <= 6) QSP_LEW_ROW(TILING13_2, 18, 6 * c + (sc - 1), 6);
            if (sc <= 18) QSP_LEW_ROW(TILING13_3, 30, 12 * c + (sc - 7), 10);
            if (sc <= 22) QSP_LEW_ROW(TILING13_4, 36, 4 * c + (sc - 19), 12);
            if (sc <= 26) {
                const int k = sc - 23;
                if (test_internal(cs, c, k, TEST13[7 * c + 6], v)) QSP_LEW_ROW(TILING13_5_1, 18, 4 * c + k, 6);
                QSP_LEW_ROW(TILING13_5_2, 30, 4 * c + k, 10);
            }
            if (sc <= 38) QSP_LEW_ROW(TILING13_3_, 30, 12 * c + (sc - 27), 10);
            if (sc <= 44) QSP_LEW_ROW(TILING13_2_, 18, 6 * c + (sc - 39), 6);
            QSP_LEW_ROW(TILING13_1_, 12, c, 4);
        }
        case 14: QSP_LEW_ROW(TILING14, 12, c, 4);
        default: *row = nullptr; return 0;
    }
#undef QSP_LEW_ROW
}

// One cell = the grid point p of its lowest corner (x fastest).  Loads the corner values; false when p is no cell or has no sign change.
struct Cell {
    int x, y, z, index;
    double v[8];
};
__device__ inline bool load_cell(const float* __restrict__ sdf, int d, int64_t p, Cell& c) {
    c.x = (int)(p % d), c.y = (int)((p / d) % d), c.z = (int)(p / ((int64_t)d * d));
    if (c.x + 1 >= d || c.y + 1 >= d || c.z + 1 >= d) return false;
    int idx = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float f = sdf[p + C_OFF[k][0] + (int64_t)C_OFF[k][1] * d + (int64_t)C_OFF[k][2] * d * d];
        c.v[k] = (double)f;
        if (f > 0.f) idx |= 1 << k;
    }
    c.index = idx;
    return idx != 0 && idx != 255;
}
// does this cell create the vertex of its edge e (is it the first of the cells around that edge in the sweep)?
__device__ inline bool owns_edge(const Cell& c, int e) {
    const int a = E_AXIS[e];
    const int px = c.x + E_LO[e][0], py = c.y + E_LO[e][1], pz = c.z + E_LO[e][2];
    // the cells around an edge along x differ in (y, z): the first one has y = max(py - 1, 0), z = max(pz - 1, 0); likewise y, z
    const int ox = a == 0 ? px : max(px - 1, 0), oy = a == 1 ? py : max(py - 1, 0), oz = a == 2 ? pz : max(pz - 1, 0);
    return ox == c.x && oy == c.y && oz == c.z;
}

// pass 1: per grid point, low 32 bits = vertices its cell creates, high 32 bits = its triangles
__global__ __launch_bounds__(256) void k_lew_count(const float* __restrict__ sdf, int d, unsigned long long* __restrict__ cnt, int64_t n_pad) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n_pad) return;
    unsigned long long out = 0;
    Cell c;
    if (p < (int64_t)d * d * d && load_cell(sdf, d, p, c)) {
        const int8_t* row;
        const int nt = cell_tiling(c.index, c.v, &row);
        unsigned seen = 0, created = 0;
        for (int j = 0; j < 3 * nt; ++j) {
            const int e = row[j];
            if (seen >> e & 1u) continue;
            seen |= 1u << e;
            if (e == 12 || owns_edge(c, e)) ++created;
        }
        out = (unsigned long long)created | ((unsigned long long)nt << 32);
    }
    cnt[p] = out;
}

// pass 2: the vertices a cell creates, numbered in the order its triangles first use them, and the edge -> vertex map
__global__ __launch_bounds__(256) void k_lew_verts(const float* __restrict__ sdf, int d, float voxel_size,
                                                   const unsigned long long* __restrict__ cnt, const unsigned long long* __restrict__ bsum,
                                                   int scan_block, float* __restrict__ vidx, float* __restrict__ verts,
                                                   int32_t* __restrict__ vmap) {
    const int64_t n = (int64_t)d * d * d;
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    Cell c;
    if (p >= n || !load_cell(sdf, d, p, c)) return;
    const int8_t* row;
    const int nt = cell_tiling(c.index, c.v, &row);
    unsigned id = (unsigned)(cnt[p] & 0xffffffffull) + (unsigned)(bsum[p / scan_block] & 0xffffffffull);
    unsigned seen = 0;
    for (int j = 0; j < 3 * nt; ++j) {
        const int e = row[j];
        if (seen >> e & 1u) continue;
        seen |= 1u << e;
        if (e != 12 && !owns_edge(c, e)) continue;
        double fx = 0, fy = 0, fz = 0, ff = 0;
        {
#pragma clang fp contract(off)
            if (e == 12) {
                for (int k = 0; k < 8; ++k) {
                    const double w = 1.0 / (EPS + fabs(c.v[k]));
                    fx += (double)C_OFF[k][0] * w;
                    fy += (double)C_OFF[k][1] * w;
                    fz += (double)C_OFF[k][2] * w;
                    ff += w;
                }
            } else {
                const int c1 = E_C1[e], c2 = E_C2[e];
                const double w1 = 1.0 / (EPS + fabs(c.v[c1])), w2 = 1.0 / (EPS + fabs(c.v[c2]));
                fx = (double)C_OFF[c1][0] * w1 + (double)C_OFF[c2][0] * w2;
                fy = (double)C_OFF[c1][1] * w1 + (double)C_OFF[c2][1] * w2;
                fz = (double)C_OFF[c1][2] * w1 + (double)C_OFF[c2][2] * w2;
                ff = w1 + w2;
                const int64_t q = p + E_LO[e][0] + (int64_t)E_LO[e][1] * d + (int64_t)E_LO[e][2] * d * d;
                vmap[(int64_t)E_AXIS[e] * n + q] = (int32_t)id;
            }
            // index coordinates as float32 in the volume's axis order (z, y, x); the float32 mesh of the C-ABI: * spacing + (-1)
            const float iz = (float)((double)c.z + fz / ff), iy = (float)((double)c.y + fy / ff), ix = (float)((double)c.x + fx / ff);
            vidx[3 * (size_t)id + 0] = iz;
            vidx[3 * (size_t)id + 1] = iy;
            vidx[3 * (size_t)id + 2] = ix;
            const float sz = iz * voxel_size, sy = iy * voxel_size, sx = ix * voxel_size;
            verts[3 * (size_t)id + 0] = sz + (-1.0f);
            verts[3 * (size_t)id + 1] = sy + (-1.0f);
            verts[3 * (size_t)id + 2] = sx + (-1.0f);
        }
        ++id;
    }
}

// pass 3: faces (corners reversed: gradient_direction = 'descent')
__global__ __launch_bounds__(256) void k_lew_faces(const float* __restrict__ sdf, int d, const unsigned long long* __restrict__ cnt,
                                                   const unsigned long long* __restrict__ bsum, int scan_block,
                                                   const int32_t* __restrict__ vmap, int32_t* __restrict__ faces) {
    const int64_t n = (int64_t)d * d * d;
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    Cell c;
    if (p >= n || !load_cell(sdf, d, p, c)) return;
    const int8_t* row;
    const int nt = cell_tiling(c.index, c.v, &row);
    const unsigned v0 = (unsigned)(cnt[p] & 0xffffffffull) + (unsigned)(bsum[p / scan_block] & 0xffffffffull);
    const size_t f0 = (size_t)(cnt[p] >> 32) + (size_t)(bsum[p / scan_block] >> 32);
    // the extra vertex's number: its rank among the vertices this cell creates
    int center = -1;
    {
        unsigned seen = 0, created = 0;
        for (int j = 0; j < 3 * nt; ++j) {
            const int e = row[j];
            if (seen >> e & 1u) continue;
            seen |= 1u << e;
            if (e == 12) { center = (int)(v0 + created); break; }
            if (owns_edge(c, e)) ++created;
        }
    }
    for (int j = 0; j < 3 * nt; ++j) {
        const int e = row[j];
        int32_t id;
        if (e == 12) id = center;
        else {
            const int64_t q = p + E_LO[e][0] + (int64_t)E_LO[e][1] * d + (int64_t)E_LO[e][2] * d * d;
            id = vmap[(int64_t)E_AXIS[e] * n + q];
        }
        const int t = j / 3, k = j - 3 * t;
        faces[3 * (f0 + t) + (2 - k)] = id;
    }
}

}  // namespace lew
}  // namespace qsp
