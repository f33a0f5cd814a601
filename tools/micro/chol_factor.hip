// Micro-benchmark of the 64x64 diagonal-block factorisation of the BA's blocked Cholesky (csrc/ba_solver.hip:factor_tile64): the
// serial chain of every block step.  One workgroup of 256 threads factorises a 64x64 SPD matrix held as 4x4 register tiles and
// produces W = L^-1; wave 3's last thread stamps the shader clock at the phase boundaries of every 4-row round.
//   hipcc --offload-arch=gfx950 -O3 -o chol_factor tools/micro/chol_factor.hip && ./chol_factor [variant]
//   variant 0: as shipped in round 2 (W carried through the rounds, 16 workgroup barriers)
//   variant 1: U only in the rounds (no W), to see what the chain costs without the inverse
//   variant 2: wave-local sub-rounds: a wave's four rounds need no workgroup barrier (4 barriers, rank-16 catch-up for the rest)
//   variant 3: one tile per thread and round: U is upper and W = L^-1 lower triangular, so a thread right of / on the diagonal
//              only ever needs its S tile and a thread left of it only its W tile (the diagonal W tiles stay the identity until
//              their own pivot round and are final after it)
//   variant 5: a fifth PANEL wave (lane = column) owns the pivot chain: in phase p it takes rows 4p..4p+3 as the updaters left them
//              one block earlier, applies block p-1 itself (8 values per lane), factorises and publishes block p -- while the four
//              updater waves are still applying block p-1 to everything below.  One barrier per phase; the chain and the rank-4
//              update overlap instead of alternating.  Same operations in the same order on every element as variant 0.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int NB = 64;

__device__ inline double rsqrt_nr(double d) {
    double r = __builtin_amdgcn_rsq(d);
    r = r * (1.5 - 0.5 * d * r * r);
    r = r * (1.5 - 0.5 * d * r * r);
    return r;
}

__device__ unsigned long long g_ts[16][4];

template <bool WITH_W>
__device__ inline void pivot_rows(double (&S)[4][4], double (&W)[4][4], double* rb, int jb, int lane, int c0) {
    const int src = (lane & 48) | jb;
    double D[4][4], rs[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = a; b < 4; ++b) D[a][b] = __shfl(S[a][b], src, 64);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        double dd = D[q][q];
        rs[q] = rsqrt_nr(dd);
#pragma unroll
        for (int b = q + 1; b < 4; ++b) D[q][b] *= rs[q];
#pragma unroll
        for (int a = q + 1; a < 4; ++a)
#pragma unroll
            for (int b = a; b < 4; ++b) D[a][b] -= D[q][a] * D[q][b];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            double sv = S[q][b], wv = W[q][b];
#pragma unroll
            for (int pp = 0; pp < q; ++pp) {
                sv -= D[pp][q] * S[pp][b];
                if (WITH_W) wv -= D[pp][q] * W[pp][b];
            }
            S[q][b] = sv * rs[q];
            rb[q * NB + c0 + b] = S[q][b];
            if (WITH_W) {
                W[q][b] = wv * rs[q];
                rb[(4 + q) * NB + c0 + b] = W[q][b];
            }
        }
}

template <bool WITH_W>
__device__ inline void rank4(double (&S)[4][4], double (&W)[4][4], const double* rb, int r0, int c0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        double ur[4], uc[4], wc[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            ur[a] = rb[q * NB + r0 + a];
            uc[a] = rb[q * NB + c0 + a];
            if (WITH_W) wc[a] = rb[(4 + q) * NB + c0 + a];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                S[a][b] -= ur[a] * uc[b];
                if (WITH_W) W[a][b] -= ur[a] * wc[b];
            }
    }
}

// variant 3 helpers: X is the thread's one live tile (S if c0 >= r0, else W)
__device__ inline void pivot_rows_sel(double (&S)[4][4], double (&W)[4][4], double* rb, int jb, int lane, int r0, int c0) {
    const int src = (lane & 48) | jb;
    double D[4][4], rs[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = a; b < 4; ++b) D[a][b] = __shfl(S[a][b], src, 64);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        double dd = D[q][q];
        rs[q] = rsqrt_nr(dd);
#pragma unroll
        for (int b = q + 1; b < 4; ++b) D[q][b] *= rs[q];
#pragma unroll
        for (int a = q + 1; a < 4; ++a)
#pragma unroll
            for (int b = a; b < 4; ++b) D[a][b] -= D[q][a] * D[q][b];
    }
    // (r0 == 4 jb here)  columns right of the diagonal: rows of U;  left of it: rows of W;  the diagonal tile: both
    const bool isU = c0 >= r0, isW = c0 <= r0;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            double sv = S[q][b], wv = W[q][b];
#pragma unroll
            for (int pp = 0; pp < q; ++pp) {
                sv -= D[pp][q] * S[pp][b];
                wv -= D[pp][q] * W[pp][b];
            }
            S[q][b] = sv * rs[q];
            W[q][b] = wv * rs[q];
            if (isU) rb[q * NB + c0 + b] = S[q][b];
            if (isW) rb[(4 + q) * NB + c0 + b] = W[q][b];
        }
}

__device__ inline void rank4_sel(double (&X)[4][4], const double* rb, int r0, int c0, bool upper) {
    const double* colbase = rb + (upper ? 0 : 4 * NB) + c0;      // U rows for the S tiles, W rows for the W tiles
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        double ur[4], xc[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            ur[a] = rb[q * NB + r0 + a];
            xc[a] = colbase[q * NB + a];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) X[a][b] -= ur[a] * xc[b];
    }
}

// variant 4: ONE live register tile X per thread (S on / right of the diagonal, W left of it) and one instruction stream; the
// diagonal threads get their W tile (Wd) in their own pivot round
__device__ inline void pivot_rows_one(double (&X)[4][4], double (&Wd)[4][4], double* rb, int jb, int lane, int r0, int c0) {
    const int src = (lane & 48) | jb;
    double D[4][4], rs[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = a; b < 4; ++b) D[a][b] = __shfl(X[a][b], src, 64);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        double dd = D[q][q];
        rs[q] = rsqrt_nr(dd);
#pragma unroll
        for (int b = q + 1; b < 4; ++b) D[q][b] *= rs[q];
#pragma unroll
        for (int a = q + 1; a < 4; ++a)
#pragma unroll
            for (int b = a; b < 4; ++b) D[a][b] -= D[q][a] * D[q][b];
    }
    const bool lower = c0 < r0, diag = c0 == r0;
    double* out = rb + (lower ? 4 * NB : 0) + c0;               // W rows go to the second half of the buffer
    double* other = rb + (lower ? 0 : 4 * NB) + c0;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            double xv = X[q][b];
#pragma unroll
            for (int pp = 0; pp < q; ++pp) xv -= D[pp][q] * X[pp][b];
            X[q][b] = xv * rs[q];
            out[q * NB + b] = X[q][b];
            if (!diag) other[q * NB + b] = 0.0;                   // U left of the diagonal / W right of it: zero
        }
    if (diag) {                                                    // W of the diagonal tile: the same row operations on the identity
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                double wv = (q == b) ? 1.0 : 0.0;
#pragma unroll
                for (int pp = 0; pp < q; ++pp) wv -= D[pp][q] * Wd[pp][b];
                Wd[q][b] = wv * rs[q];
                rb[(4 + q) * NB + c0 + b] = Wd[q][b];
            }
    }
}

template <int VAR>
__global__ __launch_bounds__(256) void k_factor(const double* A, double* Wout, double* Uout) {
    __shared__ double rowbuf[16 * 8 * NB];      // one 8 x 64 buffer per round (64 KiB)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r0 = (t >> 4) * 4, c0 = (t & 15) * 4;
    double S[4][4], W[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            S[a][b] = A[(r0 + a) * NB + c0 + b];
            W[a][b] = (r0 + a == c0 + b) ? 1.0 : 0.0;
        }
    __syncthreads();
    constexpr bool WW = VAR != 1;
    if (VAR == 4) {
        const bool upper = c0 >= r0;
        double X[4][4], Wd[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) { X[a][b] = upper ? S[a][b] : 0.0; Wd[a][b] = 0.0; }
#pragma nounroll
        for (int jb = 0; jb < 16; ++jb) {
            double* rb = rowbuf + jb * 8 * NB;
            if (t == 255) g_ts[jb][0] = __builtin_readcyclecounter();
            if ((t >> 4) == jb) pivot_rows_one(X, Wd, rb, jb, lane, r0, c0);
            if (t == 255) g_ts[jb][1] = __builtin_readcyclecounter();
            __syncthreads();
            if (t == 255) g_ts[jb][2] = __builtin_readcyclecounter();
            if (r0 > 4 * jb) rank4_sel(X, rb, r0, c0, upper);
            if (t == 255) g_ts[jb][3] = __builtin_readcyclecounter();
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                S[a][b] = upper ? X[a][b] : 0.0;
                W[a][b] = (c0 < r0) ? X[a][b] : (c0 == r0 ? Wd[a][b] : 0.0);
            }
    } else if (VAR == 3) {
        const bool upper = c0 >= r0;
#pragma nounroll
        for (int jb = 0; jb < 16; ++jb) {
            double* rb = rowbuf + jb * 8 * NB;
            if (t == 255) g_ts[jb][0] = __builtin_readcyclecounter();
            if ((t >> 4) == jb) pivot_rows_sel(S, W, rb, jb, lane, r0, c0);
            if (t == 255) g_ts[jb][1] = __builtin_readcyclecounter();
            __syncthreads();
            if (t == 255) g_ts[jb][2] = __builtin_readcyclecounter();
            if (r0 > 4 * jb) {
                if (upper) rank4_sel(S, rb, r0, c0, true);
                else rank4_sel(W, rb, r0, c0, false);
            }
            if (t == 255) g_ts[jb][3] = __builtin_readcyclecounter();
        }
    } else if (VAR == 0 || VAR == 1) {
#pragma nounroll
        for (int jb = 0; jb < 16; ++jb) {
            double* rb = rowbuf + jb * 8 * NB;
            if (t == 255) g_ts[jb][0] = __builtin_readcyclecounter();
            if ((t >> 4) == jb) pivot_rows<WW>(S, W, rb, jb, lane, c0);
            if (t == 255) g_ts[jb][1] = __builtin_readcyclecounter();
            __syncthreads();
            if (t == 255) g_ts[jb][2] = __builtin_readcyclecounter();
            if (r0 > 4 * jb) rank4<WW>(S, W, rb, r0, c0);
            if (t == 255) g_ts[jb][3] = __builtin_readcyclecounter();
        }
    } else {
#pragma nounroll
        for (int w = 0; w < 4; ++w) {
            if (t == 255) g_ts[4 * w][0] = __builtin_readcyclecounter();
            if (wave == w) {
#pragma nounroll
                for (int u = 0; u < 4; ++u) {
                    const int jb = 4 * w + u;
                    double* rb = rowbuf + jb * 8 * NB;
                    if ((lane >> 4) == u) pivot_rows<true>(S, W, rb, jb, lane, c0);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // this wave's LDS stores have landed ...
                    __builtin_amdgcn_wave_barrier();                            // ... before its other lanes read them
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    if ((lane >> 4) > u) rank4<true>(S, W, rb, r0, c0);
                }
            }
            if (t == 255) g_ts[4 * w][1] = __builtin_readcyclecounter();
            __syncthreads();
            if (t == 255) g_ts[4 * w][2] = __builtin_readcyclecounter();
            if (wave > w) {
#pragma unroll
                for (int u = 0; u < 4; ++u) rank4<true>(S, W, rowbuf + (4 * w + u) * 8 * NB, r0, c0);
            }
            if (t == 255) g_ts[4 * w][3] = __builtin_readcyclecounter();
        }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            Wout[(r0 + a) * NB + c0 + b] = W[a][b];
            Uout[(r0 + a) * NB + c0 + b] = S[a][b];
        }
}

__device__ unsigned long long g_ts5[16][4];
__device__ unsigned long long g_ts5f[16][8];
#ifdef QSP_FINE_STAMPS      // the scheduler may not move anything across a stamp; the wait makes the LDS traffic before it complete
#define QSP_STAMP(i_) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0); if (lane == 0) g_ts5f[p][i_] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); }
#else
#define QSP_STAMP(i_)
#endif
__global__ __launch_bounds__(320) void k_factor5(const double* A, double* Wout, double* Uout) {
    __shared__ double rowbuf[16 * 8 * NB];      // per phase: 4 rows of U, 4 rows of W
    __shared__ double nx[2 * 8 * NB];           // rows of the next block as the updaters left them (S: 4 x 64, W: 4 x 64)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const bool panel = wave == 4;
    const int r0 = (t >> 4) * 4, c0 = (t & 15) * 4;
    double S[4][4], W[4][4];
    if (!panel) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                S[a][b] = A[(r0 + a) * NB + c0 + b];
                W[a][b] = (r0 + a == c0 + b) ? 1.0 : 0.0;
            }
        if ((t >> 4) == 0) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) { nx[a * NB + c0 + b] = S[a][b]; nx[(4 + a) * NB + c0 + b] = W[a][b]; }
        }
    }
    __syncthreads();
#pragma nounroll
    for (int p = 0; p < 16; ++p) {
        double* rb = rowbuf + p * 8 * NB;
        const double* rbp = rowbuf + (p - 1) * 8 * NB;
        if (panel) {
            if (lane == 0) g_ts5[p][0] = __builtin_readcyclecounter();
            const double* in = nx + (p & 1) * 8 * NB;
            double s[4], w[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) { s[a] = in[a * NB + lane]; w[a] = in[(4 + a) * NB + lane]; }
            QSP_STAMP(0)
            if (p > 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const double uc = rbp[q * NB + lane], wc = rbp[(4 + q) * NB + lane];
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const double ur = rbp[q * NB + 4 * p + a];
                        s[a] -= ur * uc;
                        w[a] -= ur * wc;
                    }
                }
            }
            QSP_STAMP(1)
            double D[4][4], rs[4];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = a; b < 4; ++b) {
                    union { double d; int i[2]; } u, r;
                    u.d = s[a];
                    r.i[0] = __builtin_amdgcn_readlane(u.i[0], 4 * p + b);
                    r.i[1] = __builtin_amdgcn_readlane(u.i[1], 4 * p + b);
                    D[a][b] = r.d;
                }
            QSP_STAMP(2)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                rs[q] = rsqrt_nr(D[q][q]);
#pragma unroll
                for (int b = q + 1; b < 4; ++b) D[q][b] *= rs[q];
#pragma unroll
                for (int a = q + 1; a < 4; ++a)
#pragma unroll
                    for (int b = a; b < 4; ++b) D[a][b] -= D[q][a] * D[q][b];
            }
            QSP_STAMP(3)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                double sv = s[q], wv = w[q];
#pragma unroll
                for (int pp = 0; pp < q; ++pp) { sv -= D[pp][q] * s[pp]; wv -= D[pp][q] * w[pp]; }
                s[q] = sv * rs[q];
                w[q] = wv * rs[q];
                rb[q * NB + lane] = s[q];
                rb[(4 + q) * NB + lane] = w[q];
            }
            QSP_STAMP(4)
            if (lane == 0) g_ts5[p][1] = __builtin_readcyclecounter();
        } else {
            if (t == 255) g_ts5[p][2] = __builtin_readcyclecounter();
            if (p > 0 && r0 > 4 * (p - 1)) rank4<true>(S, W, rbp, r0, c0);
            if ((t >> 4) == p + 1) {
                double* out = nx + ((p + 1) & 1) * 8 * NB;
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) { out[a * NB + c0 + b] = S[a][b]; out[(4 + a) * NB + c0 + b] = W[a][b]; }
            }
            if (t == 255) g_ts5[p][3] = __builtin_readcyclecounter();
        }
        __syncthreads();
    }
    for (int e = t; e < NB * NB; e += 320) {
        const int q = e / NB, m = e % NB;
        Uout[e] = rowbuf[(q / 4) * 8 * NB + (q % 4) * NB + m];
        Wout[e] = rowbuf[(q / 4) * 8 * NB + (4 + q % 4) * NB + m];
    }
}

// variants 6 (R = 4) and 7 (R = 8): variant 5 with R rows per phase and the panel wave keeping the block it has just published in
// registers (its column of the R rows of U and W; the R x R diagonal values by v_readlane) instead of reading it back from LDS
template <int R>
__device__ inline void rankR(double (&S)[4][4], double (&W)[4][4], const double* rb, int r0, int c0) {
#pragma unroll
    for (int q = 0; q < R; ++q) {
        double ur[4], uc[4], wc[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            ur[a] = rb[q * NB + r0 + a];
            uc[a] = rb[q * NB + c0 + a];
            wc[a] = rb[(R + q) * NB + c0 + a];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                S[a][b] -= ur[a] * uc[b];
                W[a][b] -= ur[a] * wc[b];
            }
    }
}
__device__ inline double rdlane(double v, int l) {
    union { double d; int i[2]; } u, r;
    u.d = v;
    r.i[0] = __builtin_amdgcn_readlane(u.i[0], l);
    r.i[1] = __builtin_amdgcn_readlane(u.i[1], l);
    return r.d;
}
template <int R>
__global__ __launch_bounds__(320) void k_factorR(const double* A, double* Wout, double* Uout) {
    constexpr int NP = NB / R;
    __shared__ double rowbuf[NP * 2 * R * NB];
    __shared__ double nx[2 * 2 * R * NB];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const bool panel = wave == 4;
    const int r0 = (t >> 4) * 4, c0 = (t & 15) * 4;
    double S[4][4], W[4][4];
    double sp[R], wp[R];          // panel: the block published in the phase before
    if (!panel) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                S[a][b] = A[(r0 + a) * NB + c0 + b];
                W[a][b] = (r0 + a == c0 + b) ? 1.0 : 0.0;
            }
        if (r0 < R) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) { nx[(r0 + a) * NB + c0 + b] = S[a][b]; nx[(R + r0 + a) * NB + c0 + b] = W[a][b]; }
        }
    }
    __syncthreads();
#pragma nounroll
    for (int p = 0; p < NP; ++p) {
        double* rb = rowbuf + p * 2 * R * NB;
        const double* rbp = rowbuf + (p - 1) * 2 * R * NB;
        if (panel) {
            if (lane == 0) g_ts5[p][0] = __builtin_readcyclecounter();
            const double* in = nx + (p & 1) * 2 * R * NB;
            double s[R], w[R];
#pragma unroll
            for (int a = 0; a < R; ++a) { s[a] = in[a * NB + lane]; w[a] = in[(R + a) * NB + lane]; }
            if (p > 0) {
#pragma unroll
                for (int q = 0; q < R; ++q) {
#pragma unroll
                    for (int a = 0; a < R; ++a) {
                        const double ur = rdlane(sp[q], R * p + a);
                        s[a] -= ur * sp[q];
                        w[a] -= ur * wp[q];
                    }
                }
            }
            double D[R][R], rs[R];
#pragma unroll
            for (int a = 0; a < R; ++a)
#pragma unroll
                for (int b = a; b < R; ++b) D[a][b] = rdlane(s[a], R * p + b);
#pragma unroll
            for (int q = 0; q < R; ++q) {
                rs[q] = rsqrt_nr(D[q][q]);
#pragma unroll
                for (int b = q + 1; b < R; ++b) D[q][b] *= rs[q];
#pragma unroll
                for (int a = q + 1; a < R; ++a)
#pragma unroll
                    for (int b = a; b < R; ++b) D[a][b] -= D[q][a] * D[q][b];
            }
#pragma unroll
            for (int q = 0; q < R; ++q) {
                double sv = s[q], wv = w[q];
#pragma unroll
                for (int pp = 0; pp < q; ++pp) { sv -= D[pp][q] * s[pp]; wv -= D[pp][q] * w[pp]; }
                s[q] = sv * rs[q];
                w[q] = wv * rs[q];
                rb[q * NB + lane] = s[q];
                rb[(R + q) * NB + lane] = w[q];
                sp[q] = s[q];
                wp[q] = w[q];
            }
            if (lane == 0) g_ts5[p][1] = __builtin_readcyclecounter();
        } else {
            if (t == 255) g_ts5[p][2] = __builtin_readcyclecounter();
            if (p > 0 && r0 >= R * p) rankR<R>(S, W, rbp, r0, c0);
            if (r0 >= R * (p + 1) && r0 < R * (p + 2)) {
                double* out = nx + ((p + 1) & 1) * 2 * R * NB;
                const int rr = r0 - R * (p + 1);
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) { out[(rr + a) * NB + c0 + b] = S[a][b]; out[(R + rr + a) * NB + c0 + b] = W[a][b]; }
            }
            if (t == 255) g_ts5[p][3] = __builtin_readcyclecounter();
        }
        __syncthreads();
    }
    for (int e = t; e < NB * NB; e += 320) {
        const int q = e / NB, m = e % NB;
        Uout[e] = rowbuf[(q / R) * 2 * R * NB + (q % R) * NB + m];
        Wout[e] = rowbuf[(q / R) * 2 * R * NB + (R + q % R) * NB + m];
    }
}

// variant 8: variant 5 with the inverse one phase behind: a sixth wave carries the W rows (it needs the pivots' 4x4 factors of a
// block, which the S panel wave leaves in LDS, but nothing of it is on the S chain), and the updaters apply block p-1 to their S
// tiles and block p-2 to their W tiles.  17 phases; the S panel wave's phase is about half as long.
__global__ __launch_bounds__(384) void k_factor8(const double* A, double* Wout, double* Uout) {
    __shared__ double rbU[16 * 4 * NB];         // U rows, per block
    __shared__ double rbW[16 * 4 * NB];         // W rows, per block
    __shared__ double nxS[2 * 4 * NB], nxW[2 * 4 * NB];
    __shared__ double dbuf[2 * 16];             // per block: D[0][1], D[0][2], D[0][3], D[1][2], D[1][3], D[2][3], rs[0..3]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int r0 = (t >> 4) * 4, c0 = (t & 15) * 4;
    double S[4][4], W[4][4];
    if (wave < 4) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                S[a][b] = A[(r0 + a) * NB + c0 + b];
                W[a][b] = (r0 + a == c0 + b) ? 1.0 : 0.0;
            }
        if (r0 == 0) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) { nxS[a * NB + c0 + b] = S[a][b]; nxW[a * NB + c0 + b] = W[a][b]; }
        }
    }
    __syncthreads();
#pragma nounroll
    for (int p = 0; p <= 16; ++p) {
        if (wave == 4) {                        // S panel: block p
            if (p < 16) {
                if (lane == 0) g_ts5[p][0] = __builtin_readcyclecounter();
                const double* in = nxS + (p & 1) * 4 * NB;
                const double* up = rbU + (p - 1) * 4 * NB;
                double s[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) s[a] = in[a * NB + lane];
                if (p > 0) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const double uc = up[q * NB + lane];
#pragma unroll
                        for (int a = 0; a < 4; ++a) s[a] -= up[q * NB + 4 * p + a] * uc;
                    }
                }
                double D[4][4], rs[4];
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = a; b < 4; ++b) D[a][b] = rdlane(s[a], 4 * p + b);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    rs[q] = rsqrt_nr(D[q][q]);
#pragma unroll
                    for (int b = q + 1; b < 4; ++b) D[q][b] *= rs[q];
#pragma unroll
                    for (int a = q + 1; a < 4; ++a)
#pragma unroll
                        for (int b = a; b < 4; ++b) D[a][b] -= D[q][a] * D[q][b];
                }
                double* ub = rbU + p * 4 * NB;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    double sv = s[q];
#pragma unroll
                    for (int pp = 0; pp < q; ++pp) sv -= D[pp][q] * s[pp];
                    s[q] = sv * rs[q];
                    ub[q * NB + lane] = s[q];
                }
                if (lane == 0) {
                    double* db = dbuf + (p & 1) * 16;
                    db[0] = D[0][1]; db[1] = D[0][2]; db[2] = D[0][3]; db[3] = D[1][2]; db[4] = D[1][3]; db[5] = D[2][3];
                    db[6] = rs[0]; db[7] = rs[1]; db[8] = rs[2]; db[9] = rs[3];
                }
                if (lane == 0) g_ts5[p][1] = __builtin_readcyclecounter();
            }
        } else if (wave == 5) {                 // W panel: block p-1
            if (p >= 1) {
                const int pb = p - 1;
                const double* in = nxW + (pb & 1) * 4 * NB;
                double w[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) w[a] = in[a * NB + lane];
                if (pb > 0) {
                    const double* up = rbU + (pb - 1) * 4 * NB;
                    const double* wp = rbW + (pb - 1) * 4 * NB;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const double wc = wp[q * NB + lane];
#pragma unroll
                        for (int a = 0; a < 4; ++a) w[a] -= up[q * NB + 4 * pb + a] * wc;
                    }
                }
                const double* db = dbuf + (pb & 1) * 16;
                double D[4][4], rs[4];
                D[0][1] = db[0]; D[0][2] = db[1]; D[0][3] = db[2]; D[1][2] = db[3]; D[1][3] = db[4]; D[2][3] = db[5];
                rs[0] = db[6]; rs[1] = db[7]; rs[2] = db[8]; rs[3] = db[9];
                double* wb = rbW + pb * 4 * NB;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    double wv = w[q];
#pragma unroll
                    for (int pp = 0; pp < q; ++pp) wv -= D[pp][q] * w[pp];
                    w[q] = wv * rs[q];
                    wb[q * NB + lane] = w[q];
                }
            }
        } else {
            if (t == 255 && p < 16) g_ts5[p][2] = __builtin_readcyclecounter();
            if (p >= 1 && p <= 16 && r0 > 4 * (p - 1)) {        // S tiles: block p-1
                const double* up = rbU + (p - 1) * 4 * NB;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    double ur[4], uc[4];
#pragma unroll
                    for (int a = 0; a < 4; ++a) { ur[a] = up[q * NB + r0 + a]; uc[a] = up[q * NB + c0 + a]; }
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b = 0; b < 4; ++b) S[a][b] -= ur[a] * uc[b];
                }
            }
            if (p >= 2 && r0 > 4 * (p - 2)) {                   // W tiles: block p-2
                const double* up = rbU + (p - 2) * 4 * NB;
                const double* wp = rbW + (p - 2) * 4 * NB;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    double ur[4], wc[4];
#pragma unroll
                    for (int a = 0; a < 4; ++a) { ur[a] = up[q * NB + r0 + a]; wc[a] = wp[q * NB + c0 + a]; }
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b = 0; b < 4; ++b) W[a][b] -= ur[a] * wc[b];
                }
            }
            if (r0 == 4 * (p + 1)) {            // rows of block p+1, S through block p-1
                double* out = nxS + ((p + 1) & 1) * 4 * NB;
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) out[a * NB + c0 + b] = S[a][b];
            }
            if (p >= 1 && r0 == 4 * p) {        // rows of block p, W through block p-2
                double* out = nxW + (p & 1) * 4 * NB;
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) out[a * NB + c0 + b] = W[a][b];
            }
            if (t == 255 && p < 16) g_ts5[p][3] = __builtin_readcyclecounter();
        }
        __syncthreads();
    }
    for (int e = t; e < NB * NB; e += 384) {
        const int q = e / NB, m = e % NB;
        Uout[e] = rbU[q * NB + m];
        Wout[e] = rbW[q * NB + m];
    }
}

int main(int argc, char** argv) {
    const int var = argc > 1 ? atoi(argv[1]) : 0;
    std::vector<double> A(NB * NB), M(NB * NB);
    srand(1);
    for (auto& v : M) v = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < NB; ++i)
        for (int j = 0; j < NB; ++j) {
            double s = (i == j) ? 4.0 : 0.0;
            for (int k = 0; k < NB; ++k) s += M[i * NB + k] * M[j * NB + k];
            A[i * NB + j] = s;
        }
    double *dA, *dW, *dU;
    hipMalloc(&dA, sizeof(double) * NB * NB);
    hipMalloc(&dW, sizeof(double) * NB * NB);
    hipMalloc(&dU, sizeof(double) * NB * NB);
    hipMemcpy(dA, A.data(), sizeof(double) * NB * NB, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        for (int i = 0; i < 100; ++i) {
            if (var == 0) hipLaunchKernelGGL(k_factor<0>, dim3(1), dim3(256), 0, 0, dA, dW, dU);
            else if (var == 1) hipLaunchKernelGGL(k_factor<1>, dim3(1), dim3(256), 0, 0, dA, dW, dU);
            else if (var == 2) hipLaunchKernelGGL(k_factor<2>, dim3(1), dim3(256), 0, 0, dA, dW, dU);
            else if (var == 3) hipLaunchKernelGGL(k_factor<3>, dim3(1), dim3(256), 0, 0, dA, dW, dU);
            else if (var == 4) hipLaunchKernelGGL(k_factor<4>, dim3(1), dim3(256), 0, 0, dA, dW, dU);
            else if (var == 5) hipLaunchKernelGGL(k_factor5, dim3(1), dim3(320), 0, 0, dA, dW, dU);
            else if (var == 6) hipLaunchKernelGGL(k_factorR<4>, dim3(1), dim3(320), 0, 0, dA, dW, dU);
            else if (var == 7) hipLaunchKernelGGL(k_factorR<8>, dim3(1), dim3(320), 0, 0, dA, dW, dU);
            else hipLaunchKernelGGL(k_factor8, dim3(1), dim3(384), 0, 0, dA, dW, dU);
        }
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<double> Wh(NB * NB), Uh(NB * NB);
    hipMemcpy(Wh.data(), dW, sizeof(double) * NB * NB, hipMemcpyDeviceToHost);
    hipMemcpy(Uh.data(), dU, sizeof(double) * NB * NB, hipMemcpyDeviceToHost);
    // checks: U^T U = A (upper part of U), W U^T = I
    double eU = 0, eW = 0;
    for (int i = 0; i < NB; ++i)
        for (int j = i; j < NB; ++j) {
            double s = 0;
            for (int k = 0; k <= i; ++k) s += Uh[k * NB + i] * Uh[k * NB + j];
            eU = fmax(eU, fabs(s - A[i * NB + j]));
        }
    if (var != 1)
        for (int i = 0; i < NB; ++i)
            for (int j = 0; j < NB; ++j) {
                double s = 0;                                   // (W L)[i][j], L = U^T: L[k][j] = U[j][k], k >= j
                for (int k = j; k < NB; ++k) s += Wh[i * NB + k] * Uh[j * NB + k];
                eW = fmax(eW, fabs(s - (i == j)));
            }
    unsigned long long ts[16][4];
    hipMemcpyFromSymbol(ts, HIP_SYMBOL(g_ts), sizeof(ts));
    printf("variant %d: %.2f us per launch (100 back-to-back); |U^T U - A| %.1e, |W L - I| %.1e\n", var, 10.0 * ms, eU, eW);
    if (var >= 5) {
        hipMemcpyFromSymbol(ts, HIP_SYMBOL(g_ts5), sizeof(ts));
        unsigned long long pn = 0, up = 0;
        const int np = var == 7 ? 8 : 16;
        for (int p = 0; p < np; ++p) { pn += ts[p][1] - ts[p][0]; up += ts[p][3] - ts[p][2]; }
        printf("  cycles summed over phases: panel wave's section %llu, updater wave 3's section %llu; whole loop (panel lane 0) %llu\n", pn, up,
               ts[np - 1][1] - ts[0][0]);
        printf("  per phase, panel / updater wave 3:");
        for (int p = 0; p < np; ++p) printf(" %llu/%llu", ts[p][1] - ts[p][0], ts[p][3] - ts[p][2]);
        printf("\n");
#ifdef QSP_FINE_STAMPS
        if (var == 5) {
            unsigned long long f[16][8];
            hipMemcpyFromSymbol(f, HIP_SYMBOL(g_ts5f), sizeof(f));
            printf("  panel phases (cycles): load | update | readlanes | pivots | rows+store\n");
            for (int p = 0; p < 16; ++p)
                printf("   %2d: %llu | %llu | %llu | %llu | %llu\n", p, f[p][0] - ts[p][0], f[p][1] - f[p][0], f[p][2] - f[p][1], f[p][3] - f[p][2],
                       f[p][4] - f[p][3]);
        }
#endif
        // bit comparison with variant 0
        hipLaunchKernelGGL(k_factor<0>, dim3(1), dim3(256), 0, 0, dA, dW, dU);
        std::vector<double> W0(NB * NB), U0(NB * NB);
        hipMemcpy(W0.data(), dW, sizeof(double) * NB * NB, hipMemcpyDeviceToHost);
        hipMemcpy(U0.data(), dU, sizeof(double) * NB * NB, hipMemcpyDeviceToHost);
        int dw = 0, du = 0;
        for (int i = 0; i < NB; ++i)
            for (int j = 0; j < NB; ++j) {
                if (W0[i * NB + j] != Wh[i * NB + j]) ++dw;
                if (j >= i && U0[i * NB + j] != Uh[i * NB + j]) ++du;
            }
        printf("  elements that differ from variant 0 in any bit: W %d, U (upper) %d\n", dw, du);
        return 0;
    }
    const int step = var == 2 ? 4 : 1;
    unsigned long long tot_p = 0, tot_b = 0, tot_u = 0;
    for (int jb = 0; jb < 16; jb += step) {
        tot_p += ts[jb][1] - ts[jb][0];
        tot_b += ts[jb][2] - ts[jb][1];
        tot_u += ts[jb][3] - ts[jb][2];
    }
    printf("  thread 255, cycles summed over rounds: own pivot section %llu, waiting at the barrier %llu, update %llu; whole loop %llu\n",
           tot_p, tot_b, tot_u, ts[16 - step][3] - ts[0][0]);
    return 0;
}
