// Device-side mesh extraction: SDF volume on the reference's voxel grid (reconstruct/utils.py:98-117) decoded with the MLP
// tile kernel, then marching cubes on the GPU (replaces skimage.measure.marching_cubes_lewiner called from
// reconstruct/utils.py:120-141; MeshExtractor.extract_mesh_from_code, reconstruct/optimizer.py:284-304).
// Included at the end of sdf_refine.hip (same translation unit: it launches k_decode and reads qsp_decoder).
//
// Marching cubes: method 0 (default, round 4) is Lewiner's, exactly as scikit-image's marching_cubes_lewiner runs it -- mesh_lewiner.hpp,
// pinned by scikit-image's own output.  Method 1 is the triangulation rounds 2-3 shipped when the dependency had not been found in the
// image, generated from first principles:
//   * a vertex on every grid edge whose end points differ in sign (inside = sdf < 0), at the linear zero crossing --
//     the same vertex set every marching-cubes variant has, shared between the cells around the edge;
//   * per cell, each cube face with 2 crossings contributes one segment, a face with 4 (ambiguous) two segments that
//     cut off its inside corners separately -- a rule that depends on the face's own corner signs only, so neighbouring
//     cells always agree and the surface is watertight; the segments close into loops, each loop is oriented so that
//     the right-hand normal points to increasing sdf (outwards) and fan-triangulated from the first vertex whose fan has
//     no diagonal inside a cube face (such a fan exists for every loop of every case).
// Where Lewiner differs from method 1: the diagonals inside a cell's polygons, interior ambiguity tests and the extra centre vertex
// of a few of the 33 cases, face order, vertex order.  Both share the scan kernels and the buffers of this file.
#pragma once

namespace qsp {
namespace mc {

constexpr int TMAX = 8;   // triangles per cell the generated table may need (asserted at start-up)

struct Tables {
    int8_t ntri[256];
    int8_t tri[256][TMAX * 3];   // cube-edge ids 0..11: id = 4 * axis + (u + 2 v), (u, v) offsets along the other two axes
};

inline void edge_ends(int e, int& c0, int& c1) {
    const int a = e >> 2, u = e & 1, v = (e >> 1) & 1;
    int base;
    if (a == 0) base = (u << 1) | (v << 2);
    else if (a == 1) base = u | (v << 2);
    else base = u | (v << 1);
    c0 = base;
    c1 = base | (1 << a);
}

inline int edge_between(int ca, int cb) {
    for (int e = 0; e < 12; ++e) {
        int c0, c1;
        edge_ends(e, c0, c1);
        if ((c0 == ca && c1 == cb) || (c0 == cb && c1 == ca)) return e;
    }
    return -1;
}

// bit (2 * axis + side) set when both end points of cube edge e lie on that cube face
inline int edge_face_mask(int e) {
    int c0, c1, m = 0;
    edge_ends(e, c0, c1);
    for (int a = 0; a < 3; ++a)
        for (int sd = 0; sd < 2; ++sd)
            if (((c0 >> a) & 1) == sd && ((c1 >> a) & 1) == sd) m |= 1 << (2 * a + sd);
    return m;
}

inline bool build_tables(Tables& T) {
    for (int cs = 0; cs < 256; ++cs) {
        int nb[12][2], deg[12];
        for (int e = 0; e < 12; ++e) { deg[e] = 0; nb[e][0] = nb[e][1] = -1; }
        auto link = [&](int ea, int eb) {
            if (deg[ea] < 2) nb[ea][deg[ea]] = eb;
            if (deg[eb] < 2) nb[eb][deg[eb]] = ea;
            deg[ea]++; deg[eb]++;
        };
        for (int a = 0; a < 3; ++a)
            for (int sd = 0; sd < 2; ++sd) {
                const int b = (a + 1) % 3, c = (a + 2) % 3;
                const int ob[4] = {0, 1, 1, 0}, oc[4] = {0, 0, 1, 1};
                int q[4], in[4], fe[4];
                for (int k = 0; k < 4; ++k) {
                    q[k] = (sd << a) | (ob[k] << b) | (oc[k] << c);
                    in[k] = (cs >> q[k]) & 1;
                }
                int ncross = 0, cross[4];
                for (int k = 0; k < 4; ++k) {
                    fe[k] = edge_between(q[k], q[(k + 1) & 3]);
                    if (in[k] != in[(k + 1) & 3]) cross[ncross++] = k;
                }
                if (ncross == 2) link(fe[cross[0]], fe[cross[1]]);
                else if (ncross == 4)
                    for (int k = 0; k < 4; ++k)
                        if (in[k]) link(fe[(k + 3) & 3], fe[k]);      // the two face edges meeting at inside corner k
            }
        bool used[12] = {};
        int nt = 0;
        for (int e0 = 0; e0 < 12; ++e0) {
            if (deg[e0] == 0 || used[e0]) continue;
            if (deg[e0] != 2) return false;
            int loop[12], n = 0, prev = -1, cur = e0;
            do {
                loop[n++] = cur;
                used[cur] = true;
                int nx = nb[cur][0], ny = nb[cur][1];
                int next;
                if (prev < 0) next = nx < ny ? nx : ny;
                else next = (nx == prev) ? ny : nx;
                if (nx == ny) next = nx;          // two-edge loops cannot occur on a cube, kept for safety
                prev = cur;
                cur = next;
            } while (cur != e0 && n < 12);
            if (n < 3) return false;
            // orientation: Newell normal of the mid-point polygon against the inside -> outside direction
            double nrm[3] = {0, 0, 0}, g[3] = {0, 0, 0}, P[12][3];
            for (int k = 0; k < n; ++k) {
                int c0, c1;
                edge_ends(loop[k], c0, c1);
                for (int ax = 0; ax < 3; ++ax) {
                    const double p0 = (c0 >> ax) & 1, p1 = (c1 >> ax) & 1;
                    P[k][ax] = 0.5 * (p0 + p1);
                    const int in0 = (cs >> c0) & 1;
                    g[ax] += in0 ? (p1 - p0) : (p0 - p1);
                }
            }
            for (int k = 0; k < n; ++k) {
                const double* p = P[k];
                const double* qn = P[(k + 1) % n];
                nrm[0] += (p[1] - qn[1]) * (p[2] + qn[2]);
                nrm[1] += (p[2] - qn[2]) * (p[0] + qn[0]);
                nrm[2] += (p[0] - qn[0]) * (p[1] + qn[1]);
            }
            const double dot = nrm[0] * g[0] + nrm[1] * g[1] + nrm[2] * g[2];
            if (dot == 0) return false;
            if (dot < 0)
                for (int lo = 1, hi = n - 1; lo < hi; ++lo, --hi) { const int tmp = loop[lo]; loop[lo] = loop[hi]; loop[hi] = tmp; }
            // fan apex: the first loop vertex none of whose fan diagonals joins two edges of one cube face -- such a
            // diagonal would lie IN that face, where the neighbouring cell could pick the same one (a non-manifold flap)
            int apex = -1;
            for (int s0 = 0; s0 < n && apex < 0; ++s0) {
                bool ok = true;
                for (int k = 2; k + 1 < n; ++k)
                    if (edge_face_mask(loop[s0]) & edge_face_mask(loop[(s0 + k) % n])) ok = false;
                if (ok) apex = s0;
            }
            if (apex < 0) return false;
            for (int k = 1; k + 1 < n; ++k) {
                if (nt >= TMAX) return false;
                T.tri[cs][3 * nt] = (int8_t)loop[apex];
                T.tri[cs][3 * nt + 1] = (int8_t)loop[(apex + k) % n];
                T.tri[cs][3 * nt + 2] = (int8_t)loop[(apex + k + 1) % n];
                ++nt;
            }
        }
        T.ntri[cs] = (int8_t)nt;
        for (int k = 3 * nt; k < TMAX * 3; ++k) T.tri[cs][k] = -1;
    }
    return true;
}

constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_BLOCK = 256 * SCAN_ITEMS;

// per grid point: bits 0..2 = its three owned edges (+axis 0/1/2) cross the surface; packed counts: low 32 bits
// vertices owned by the point, high 32 bits triangles of the cell whose lowest corner it is
__global__ __launch_bounds__(256) void k_mc_flags(const float* __restrict__ sdf, int d, const Tables* __restrict__ T,
                                                  uint8_t* __restrict__ flags, unsigned long long* __restrict__ cnt, int64_t n_pad) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n_pad) return;
    const int64_t n = (int64_t)d * d * d;
    if (p >= n) { cnt[p] = 0; return; }
    const int i2 = (int)(p % d), i1 = (int)((p / d) % d), i0 = (int)(p / ((int64_t)d * d));
    const int64_t st[3] = {(int64_t)d * d, d, 1};
    const int idx[3] = {i0, i1, i2};
    const bool in0 = sdf[p] < 0.f;
    unsigned f = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a)
        if (idx[a] + 1 < d && ((sdf[p + st[a]] < 0.f) != in0)) f |= 1u << a;
    unsigned nt = 0;
    if (i0 + 1 < d && i1 + 1 < d && i2 + 1 < d) {
        unsigned cs = 0;
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (sdf[p + (c & 1) * st[0] + ((c >> 1) & 1) * st[1] + ((c >> 2) & 1) * st[2]] < 0.f) cs |= 1u << c;
        nt = (unsigned)T->ntri[cs];
    }
    flags[p] = (uint8_t)f;
    cnt[p] = (unsigned long long)__popc(f) | ((unsigned long long)nt << 32);
}

// exclusive scan inside blocks of SCAN_BLOCK items (in place) + block totals
__global__ __launch_bounds__(256) void k_mc_scan_blocks(unsigned long long* __restrict__ cnt, unsigned long long* __restrict__ bsum) {
    __shared__ unsigned long long wsum[4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    unsigned long long* base = cnt + (size_t)blockIdx.x * SCAN_BLOCK + (size_t)t * SCAN_ITEMS;
    unsigned long long v[SCAN_ITEMS], tot = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) { v[i] = base[i]; tot += v[i]; }
    unsigned long long inc = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long up = __shfl_up(inc, o, 64);
        if (lane >= o) inc += up;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    unsigned long long off = inc - tot;
    for (int w = 0; w < wave; ++w) off += wsum[w];
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) { base[i] = off; off += v[i]; }
    if (t == 255) bsum[blockIdx.x] = off;
}

// exclusive scan of the block totals (<= 1024) by one workgroup; total[0] = grand total
__global__ __launch_bounds__(1024) void k_mc_scan_top(unsigned long long* __restrict__ bsum, int nb, unsigned long long* __restrict__ total) {
    __shared__ unsigned long long wsum[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const unsigned long long v = t < nb ? bsum[t] : 0;
    unsigned long long inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long up = __shfl_up(inc, o, 64);
        if (lane >= o) inc += up;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    unsigned long long off = inc - v;
    for (int w = 0; w < wave; ++w) off += wsum[w];
    if (t < nb) bsum[t] = off;
    if (t == 1023) total[0] = off + v;
}

__device__ inline unsigned vertex_id(const uint8_t* flags, const unsigned long long* cnt, const unsigned long long* bsum, int64_t q, int a) {
    const unsigned base = (unsigned)(cnt[q] & 0xffffffffull) + (unsigned)(bsum[q / SCAN_BLOCK] & 0xffffffffull);
    return base + __popc((unsigned)flags[q] & ((1u << a) - 1u));
}

__global__ __launch_bounds__(256) void k_mc_emit(const float* __restrict__ sdf, int d, float voxel_size, const Tables* __restrict__ T,
                                                 const uint8_t* __restrict__ flags, const unsigned long long* __restrict__ cnt,
                                                 const unsigned long long* __restrict__ bsum, float* __restrict__ verts,
                                                 int32_t* __restrict__ faces) {
    const int64_t n = (int64_t)d * d * d;
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const int i2 = (int)(p % d), i1 = (int)((p / d) % d), i0 = (int)(p / ((int64_t)d * d));
    const int64_t st[3] = {(int64_t)d * d, d, 1};
    const int idx[3] = {i0, i1, i2};
    const unsigned f = flags[p];
    const float v0 = sdf[p];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (!(f & (1u << a))) continue;
        const float v1 = sdf[p + st[a]];
        const unsigned vid = vertex_id(flags, cnt, bsum, p, a);
        {
#pragma clang fp contract(off)      // separately rounded multiply and add, like the numpy restatement (bit-exact vertices)
            const float t = v0 / (v0 - v1);
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                const float ci = (ax == a) ? (float)idx[ax] + t : (float)idx[ax];
                const float scaled = ci * voxel_size;
                verts[3 * (size_t)vid + ax] = scaled + (-1.0f);
            }
        }
    }
    if (i0 + 1 < d && i1 + 1 < d && i2 + 1 < d) {
        unsigned cs = 0;
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (sdf[p + (c & 1) * st[0] + ((c >> 1) & 1) * st[1] + ((c >> 2) & 1) * st[2]] < 0.f) cs |= 1u << c;
        const int nt = T->ntri[cs];
        if (nt) {
            const size_t f0 = (size_t)(cnt[p] >> 32) + (size_t)(bsum[p / SCAN_BLOCK] >> 32);
            for (int k = 0; k < 3 * nt; ++k) {
                const int e = T->tri[cs][k];
                const int a = e >> 2, u = e & 1, v = (e >> 1) & 1;
                int o[3];
                if (a == 0) { o[0] = 0; o[1] = u; o[2] = v; }
                else if (a == 1) { o[0] = u; o[1] = 0; o[2] = v; }
                else { o[0] = u; o[1] = v; o[2] = 0; }
                const int64_t q = p + o[0] * st[0] + o[1] * st[1] + o[2] * st[2];
                faces[3 * f0 + k] = (int32_t)vertex_id(flags, cnt, bsum, q, a);
            }
        }
    }
}

}  // namespace mc
}  // namespace qsp

#include "mesh_lewiner.hpp"

struct qsp_mesh_extractor {
    qsp_decoder* dec = nullptr;
    int marched_method = 0; // of the mesh that is resident
    int method = 0;         // 0: Lewiner's marching cubes, what the reference calls (mesh_lewiner.hpp); 1: the face-consistent table of rounds 2-3
    float* vidx = nullptr;  // Lewiner: the vertices as float32 index coordinates (the float64 mesh of the reference is these x spacing - 1)
    int32_t* vmap = nullptr;   // Lewiner: (axis, grid point) -> vertex number
    int dim = 0;
    int64_t n = 0, n_pad = 0;
    int nb = 0;
    float voxel_size = 0;
    float *xyz = nullptr, *sdf = nullptr, *code = nullptr, *verts = nullptr;
    int32_t* faces = nullptr;
    uint8_t* flags = nullptr;
    unsigned long long *cnt = nullptr, *bsum = nullptr, *total = nullptr;
    qsp::mc::Tables* tables = nullptr;
    int64_t n_verts = 0, n_faces = 0, cap_verts = 0, cap_faces = 0;
    bool have_volume = false;
    int device = 0;         // of the decoder, cached: destroy must not touch a decoder that may already be gone
};

extern "C" void qsp_mesh_extractor_destroy(qsp_mesh_extractor* m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    void* ptrs[] = {m->xyz, m->sdf, m->code, m->verts, m->faces, m->flags, m->cnt, m->bsum, m->total, m->tables, m->vidx, m->vmap};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    delete m;
    (void)hipGetLastError();   // errors are ignored here; do not leave one behind for the next call's launch check
}

extern "C" int qsp_mesh_extractor_create(qsp_decoder* dec, int32_t voxels_dim, const float* voxel_points,
                                         qsp_mesh_extractor** out) {
    using namespace qsp;
    if (!dec || !out || !voxel_points) return qsp_fail(QSP_ERR_INVALID, "qsp_mesh_extractor_create: null argument");
    if (voxels_dim < 2 || voxels_dim > 128) return qsp_fail(QSP_ERR_UNSUPPORTED, "qsp_mesh_extractor_create: 2 <= voxels_dim <= 128");
    static mc::Tables host_tables;
    static const bool tables_ok = mc::build_tables(host_tables);
    if (!tables_ok) return qsp_fail(QSP_ERR_DEVICE, "marching-cubes table generation failed");
    QSP_HIP(hipSetDevice(dec->device));
    qsp_mesh_extractor* m = new qsp_mesh_extractor();
    m->dec = dec;
    m->device = dec->device;
    m->dim = voxels_dim;
    m->n = (int64_t)voxels_dim * voxels_dim * voxels_dim;
    m->nb = (int)((m->n + mc::SCAN_BLOCK - 1) / mc::SCAN_BLOCK);
    m->n_pad = (int64_t)m->nb * mc::SCAN_BLOCK;
    m->voxel_size = (float)(2.0 / (voxels_dim - 1));
    int rc = QSP_OK;
#define MAL(field, bytes)                                                                     \
    if (!rc) {                                                                                \
        hipError_t e_ = hipMalloc((void**)&m->field, (bytes));                                \
        if (e_ != hipSuccess) rc = qsp_fail(QSP_ERR_DEVICE, hipGetErrorString(e_));           \
    }
    MAL(xyz, sizeof(float) * 3 * m->n);
    MAL(sdf, sizeof(float) * m->n);
    MAL(code, sizeof(float) * CODE_LEN);
    MAL(flags, m->n_pad);
    MAL(cnt, sizeof(unsigned long long) * m->n_pad);
    MAL(bsum, sizeof(unsigned long long) * 1024);
    MAL(total, sizeof(unsigned long long));
    MAL(tables, sizeof(mc::Tables));
    MAL(vmap, sizeof(int32_t) * 3 * m->n);
#undef MAL
    if (!rc) {
        hipError_t e = hipMemcpy(m->xyz, voxel_points, sizeof(float) * 3 * m->n, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(m->tables, &host_tables, sizeof(mc::Tables), hipMemcpyHostToDevice);
        if (e != hipSuccess) rc = qsp_fail(QSP_ERR_DEVICE, hipGetErrorString(e));
    }
    if (rc) {
        qsp_mesh_extractor_destroy(m);
        return rc;
    }
    *out = m;
    return QSP_OK;
}

static int mesh_march(qsp_mesh_extractor* m, int64_t* n_verts, int64_t* n_faces) {
    using namespace qsp;
    hipStream_t s = m->dec->stream;
    const int g = (int)(m->n_pad / 256);
    const bool lewiner = m->method == 0;
    if (lewiner) hipLaunchKernelGGL(lew::k_lew_count, dim3(g), dim3(256), 0, s, m->sdf, m->dim, m->cnt, m->n_pad);
    else hipLaunchKernelGGL(mc::k_mc_flags, dim3(g), dim3(256), 0, s, m->sdf, m->dim, m->tables, m->flags, m->cnt, m->n_pad);
    hipLaunchKernelGGL(mc::k_mc_scan_blocks, dim3(m->nb), dim3(256), 0, s, m->cnt, m->bsum);
    hipLaunchKernelGGL(mc::k_mc_scan_top, dim3(1), dim3(1024), 0, s, m->bsum, m->nb, m->total);
    unsigned long long tot = 0;
    QSP_HIP(hipMemcpyAsync(&tot, m->total, sizeof(tot), hipMemcpyDeviceToHost, s));
    QSP_HIP(hipStreamSynchronize(s));
    m->n_verts = (int64_t)(tot & 0xffffffffull);
    m->n_faces = (int64_t)(tot >> 32);
    if (m->n_verts > m->cap_verts) {
        if (m->verts) (void)hipFree(m->verts);
        if (m->vidx) (void)hipFree(m->vidx);
        m->verts = m->vidx = nullptr;
        m->cap_verts = m->n_verts + m->n_verts / 4 + 1024;
        QSP_HIP(hipMalloc((void**)&m->verts, sizeof(float) * 3 * m->cap_verts));
        QSP_HIP(hipMalloc((void**)&m->vidx, sizeof(float) * 3 * m->cap_verts));
    }
    if (m->n_faces > m->cap_faces) {
        if (m->faces) (void)hipFree(m->faces);
        m->faces = nullptr;
        m->cap_faces = m->n_faces + m->n_faces / 4 + 1024;
        QSP_HIP(hipMalloc((void**)&m->faces, sizeof(int32_t) * 3 * m->cap_faces));
    }
    if (m->n_verts && lewiner) {
        const int gp = (int)((m->n + 255) / 256);
        hipLaunchKernelGGL(lew::k_lew_verts, dim3(gp), dim3(256), 0, s, m->sdf, m->dim, m->voxel_size, m->cnt, m->bsum, mc::SCAN_BLOCK,
                           m->vidx, m->verts, m->vmap);
        hipLaunchKernelGGL(lew::k_lew_faces, dim3(gp), dim3(256), 0, s, m->sdf, m->dim, m->cnt, m->bsum, mc::SCAN_BLOCK, m->vmap, m->faces);
    } else if (m->n_verts)
        hipLaunchKernelGGL(mc::k_mc_emit, dim3((int)((m->n + 255) / 256)), dim3(256), 0, s, m->sdf, m->dim, m->voxel_size, m->tables,
                           m->flags, m->cnt, m->bsum, m->verts, m->faces);
    m->marched_method = m->method;
    QSP_HIP(hipGetLastError());
    QSP_HIP(hipStreamSynchronize(s));
    m->have_volume = true;
    if (n_verts) *n_verts = m->n_verts;
    if (n_faces) *n_faces = m->n_faces;
    return QSP_OK;
}

static int mesh_decode(qsp_mesh_extractor* m, const float* code, bool* hit) {
    using namespace qsp;
    hipStream_t s = m->dec->stream;
    float code64[CODE_LEN] = {};                       // `code` holds the decoder's code_len entries
    memcpy(code64, code, sizeof(float) * m->dec->code_len);
    QSP_HIP(hipMemcpyAsync(m->code, code64, CODE_LEN * sizeof(float), hipMemcpyHostToDevice, s));
    QSP_HIP(hipStreamSynchronize(s));                  // (code64 lives on this stack frame)
    const int64_t tiles = (m->n + TILE_P - 1) / TILE_P;
    const int grid = (int)std::min<int64_t>(tiles, 4096);
    if (m->dec->fwd_bf3 == 2)
        if (m->dec->P.narrow) hipLaunchKernelGGL((k_decode_h2<false, true>), dim3(grid), dim3(H2_THREADS), sizeof(MlpSmem), s, m->code, m->xyz, m->n,
                                                 m->dec->Pd, m->sdf, (float*)nullptr);
        else hipLaunchKernelGGL((k_decode_h2<false, false>), dim3(grid), dim3(H2_THREADS), sizeof(MlpSmem), s, m->code, m->xyz, m->n, m->dec->Pd,
                                m->sdf, (float*)nullptr);
    else if (m->dec->fwd_bf3)
        hipLaunchKernelGGL((k_decode<false, true>), dim3(grid), dim3(MLP_THREADS), sizeof(MlpSmem), s, m->code, m->xyz, m->n,
                           m->dec->Pd, m->sdf, (float*)nullptr);
    else
        hipLaunchKernelGGL(k_decode<false>, dim3(grid), dim3(MLP_THREADS), sizeof(MlpSmem), s, m->code, m->xyz, m->n, m->dec->Pd,
                           m->sdf, (float*)nullptr);
    QSP_HIP(hipGetLastError());
    QSP_HIP(hipStreamSynchronize(s));
    *hit = range_hit(m->dec);
    return QSP_OK;
}

extern "C" int qsp_mesh_extract(qsp_mesh_extractor* m, const float* code, int64_t* n_verts, int64_t* n_faces) {
    std::unique_lock<std::recursive_mutex> lk_d;
    if (m && m->dec) lk_d = std::unique_lock<std::recursive_mutex>(m->dec->mu);
    if (!m || !code) return qsp_fail(QSP_ERR_INVALID, "qsp_mesh_extract: null argument");
    QSP_HIP(hipSetDevice(m->dec->device));
    bool hit = false;
    int rc = mesh_decode(m, code, &hit);
    if (rc) return rc;
    if (hit) {      // a value of the grid decode left fp16's range: the volume is decoded again on the f32 pipe (or the call fails)
        if (!range_should_fall_back(m->dec)) return range_error();
        F32Override f32(m->dec);
        m->dec->n_range_fallbacks++;
        rc = mesh_decode(m, code, &hit);
        if (rc) return rc;
    }
    return mesh_march(m, n_verts, n_faces);
}

extern "C" int qsp_mesh_from_volume(qsp_mesh_extractor* m, const float* sdf_volume, int64_t* n_verts, int64_t* n_faces) {
    std::unique_lock<std::recursive_mutex> lk_d;
    if (m && m->dec) lk_d = std::unique_lock<std::recursive_mutex>(m->dec->mu);
    if (!m || !sdf_volume) return qsp_fail(QSP_ERR_INVALID, "qsp_mesh_from_volume: null argument");
    QSP_HIP(hipSetDevice(m->dec->device));
    QSP_HIP(hipMemcpyAsync(m->sdf, sdf_volume, sizeof(float) * m->n, hipMemcpyHostToDevice, m->dec->stream));
    return mesh_march(m, n_verts, n_faces);
}

extern "C" int qsp_mesh_fetch(qsp_mesh_extractor* m, float* verts, int32_t* faces, float* sdf_volume) {
    if (!m) return qsp_fail(QSP_ERR_INVALID, "qsp_mesh_fetch: null extractor");
    if (!m->have_volume) return qsp_fail(QSP_ERR_INVALID, "qsp_mesh_fetch: nothing extracted yet");
    QSP_HIP(hipSetDevice(m->dec->device));
    hipStream_t s = m->dec->stream;
    if (verts && m->n_verts) QSP_HIP(hipMemcpyAsync(verts, m->verts, sizeof(float) * 3 * m->n_verts, hipMemcpyDeviceToHost, s));
    if (faces && m->n_faces) QSP_HIP(hipMemcpyAsync(faces, m->faces, sizeof(int32_t) * 3 * m->n_faces, hipMemcpyDeviceToHost, s));
    if (sdf_volume) QSP_HIP(hipMemcpyAsync(sdf_volume, m->sdf, sizeof(float) * m->n, hipMemcpyDeviceToHost, s));
    QSP_HIP(hipStreamSynchronize(s));
    return QSP_OK;
}

extern "C" int qsp_mesh_extractor_set_method(qsp_mesh_extractor* m, int32_t method) {
    if (!m) return qsp_fail(QSP_ERR_INVALID, "qsp_mesh_extractor_set_method: null extractor");
    if (method != 0 && method != 1) return qsp_fail(QSP_ERR_INVALID, "mesh method: 0 (Lewiner's marching cubes, the reference's) or 1 (face-consistent table)");
    m->method = method;
    return QSP_OK;
}

// the vertices as the reference holds them: float64 = float32 index coordinate x (2 / (n - 1)) + (-1) (skimage multiplies its
// float32 vertices by the float64 spacing, reconstruct/utils.py:131-139 adds the origin)
extern "C" int qsp_mesh_fetch_f64(qsp_mesh_extractor* m, double* verts) {
    if (!m || !verts) return qsp_fail(QSP_ERR_INVALID, "qsp_mesh_fetch_f64: null argument");
    if (!m->have_volume) return qsp_fail(QSP_ERR_INVALID, "qsp_mesh_fetch_f64: nothing extracted yet");
    if (!m->n_verts) return QSP_OK;
    QSP_HIP(hipSetDevice(m->dec->device));
    std::vector<float> tmp((size_t)3 * m->n_verts);
    const bool lewiner = m->marched_method == 0;
    QSP_HIP(hipMemcpyAsync(tmp.data(), lewiner ? m->vidx : m->verts, sizeof(float) * tmp.size(), hipMemcpyDeviceToHost, m->dec->stream));
    QSP_HIP(hipStreamSynchronize(m->dec->stream));
    const double spacing = 2.0 / (double)(m->dim - 1);
    for (size_t i = 0; i < tmp.size(); ++i) verts[i] = lewiner ? (double)tmp[i] * spacing + (-1.0) : (double)tmp[i];
    return QSP_OK;
}

extern "C" int qsp_mc_tables(int8_t* ntri /*256*/, int8_t* tri /*256 x 24*/) {
    static qsp::mc::Tables t;
    static const bool ok = qsp::mc::build_tables(t);
    if (!ok) return qsp_fail(QSP_ERR_DEVICE, "marching-cubes table generation failed");
    if (ntri) memcpy(ntri, t.ntri, 256);
    if (tri) memcpy(tri, t.tri, 256 * qsp::mc::TMAX * 3);
    return QSP_OK;
}
