// LayerNorm forward / backward, one wave per token row, row cached in registers (float4 per lane).
// Replaces nn.LayerNorm in Block (vit_models/dynamic_vit.py:245,250,266,268; eps 1e-6 at :678), the final norm
// (:993) and the predictor's LayerNorms (:491-531, default eps 1e-5).  HBM-bound: one read + one write per
// element in forward; backward reads x and dy once, writes dx once, and produces deterministic per-block
// partial sums for dweight / dbias that a second tiny kernel folds.
//
// Rows may be addressed through a RowMap so that the predictor can read x[:, 1:] of a [B, n, D] buffer in place
// (dynamic_vit.py:855) and its backward can add into rows 1.. of the gradient buffer.
#include "d2s_common.h"
#include <cstdlib>

namespace {

struct RowMap {  // logical row r -> element offset
    long rows_per_group, group_stride, row_stride, offset;
};
__device__ __forceinline__ long map_row(const RowMap& m, long r) {
    const long g = r / m.rows_per_group, t = r - g * m.rows_per_group;
    return g * m.group_stride + m.offset + t * m.row_stride;
}

typedef __bf16 ln_bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_bf16x4(__bf16* __restrict__ p, const f32x4& o) {      // the bf16 copy a following bf16-mode GEMM reads
    ln_bf16x4 h;
#pragma unroll
    for (int j = 0; j < 4; ++j) h[j] = (__bf16)o[j];
    *reinterpret_cast<ln_bf16x4*>(p) = h;
}

template <int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, RowMap xm, const float* __restrict__ w,
                                                     const float* __restrict__ b, float* __restrict__ y, __bf16* __restrict__ y16,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                     long rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + map_row(xm, row);
    const int nvec = D >> 2;
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + i * 64;
        if (c < nvec) {
            v[i] = *reinterpret_cast<const f32x4*>(xr + c * 4);
            s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        } else {
            v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + i * 64;
        if (c < nvec) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = v[i][j] - mean;
                q += d * d;
            }
        }
    }
    const float var = wave_sum(q) / (float)D;
    const float rstd = 1.0f / sqrtf(var + eps);
    float* yr = y + row * D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + i * 64;
        if (c < nvec) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(w + c * 4);
            const f32x4 bv = *reinterpret_cast<const f32x4*>(b + c * 4);
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mean) * rstd * wv[j] + bv[j];
            if (y) *reinterpret_cast<f32x4*>(yr + c * 4) = o;
            if (y16) store_bf16x4(y16 + row * D + c * 4, o);
        }
    }
    if (lane == 0) {
        if (mean_out) mean_out[row] = mean;
        if (rstd_out) rstd_out[row] = rstd;
    }
}

// D = 384 (every DeiT-S block LayerNorm): a row is 96 float4, i.e. 1.5 per lane - half the wave idles on the second load.  Two rows
// per wave are 192 float4 = 3 per lane, all lanes busy and 3 loads in flight per lane.  Lane l holds vectors c = l, 64 + l, 128 + l of the
// concatenated pair; c < 96 belongs to the first row.  The reductions are segmented (one sum per row), same count as before.
__global__ __launch_bounds__(256) void ln_fwd_pair96_kernel(const float* __restrict__ x, RowMap xm, const float* __restrict__ w,
                                                            const float* __restrict__ b, float* __restrict__ y, __bf16* __restrict__ y16,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                            long rows, float eps) {
    constexpr int NVEC = 96, D = 384;
    const int lane = threadIdx.x & 63;
    const long pair = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long r0 = pair * 2;
    if (r0 >= rows) return;
    const bool has1 = r0 + 1 < rows;
    const float* xr0 = x + map_row(xm, r0);
    const float* xr1 = has1 ? x + map_row(xm, r0 + 1) : xr0;
    f32x4 v[3];
    int col[3];
    bool second[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = lane + 64 * i;
        second[i] = c >= NVEC;
        col[i] = second[i] ? c - NVEC : c;
        v[i] = *reinterpret_cast<const f32x4*>((second[i] ? xr1 : xr0) + col[i] * 4);
    }
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float t = (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        if (second[i]) s1 += t; else s0 += t;
    }
    const float mean0 = wave_sum(s0) / (float)D, mean1 = wave_sum(s1) / (float)D;
    float q0 = 0.f, q1 = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float m = second[i] ? mean1 : mean0;
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float d = v[i][j] - m; t += d * d; }
        if (second[i]) q1 += t; else q0 += t;
    }
    const float rstd0 = 1.0f / sqrtf(wave_sum(q0) / (float)D + eps), rstd1 = 1.0f / sqrtf(wave_sum(q1) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (second[i] && !has1) continue;
        const float m = second[i] ? mean1 : mean0, r = second[i] ? rstd1 : rstd0;
        const f32x4 wv = *reinterpret_cast<const f32x4*>(w + col[i] * 4);
        const f32x4 bv = *reinterpret_cast<const f32x4*>(b + col[i] * 4);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - m) * r * wv[j] + bv[j];
        const long off = (r0 + (second[i] ? 1 : 0)) * D + col[i] * 4;
        if (y) *reinterpret_cast<f32x4*>(y + off) = o;
        if (y16) store_bf16x4(y16 + off, o);
    }
    if (lane == 0) {
        if (mean_out) { mean_out[r0] = mean0; if (has1) mean_out[r0 + 1] = mean1; }
        if (rstd_out) { rstd_out[r0] = rstd0; if (has1) rstd_out[r0 + 1] = rstd1; }
    }
}

// Backward.  Each block owns a contiguous chunk of rows; every wave walks its share and accumulates dweight /
// dbias in registers; partials land in part[block][2][D].
template <int NV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ x, RowMap xm, const float* __restrict__ dy,
                                                     const float* __restrict__ w, const float* __restrict__ mean_in,
                                                     const float* __restrict__ rstd_in, float* __restrict__ dx, RowMap dxm,
                                                     const float* __restrict__ add_src, float* __restrict__ part,
                                                     long rows, int D, long rows_per_block, int relu_mask, __bf16* __restrict__ dx16) {
    extern __shared__ __attribute__((aligned(16))) float red[];  // [3][2][D]  (waves 1..3)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nvec = D >> 2;
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = min(rows, r0 + rows_per_block);
    f32x4 wv[NV], dw[NV], db[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + i * 64;
        wv[i] = c < nvec ? *reinterpret_cast<const f32x4*>(w + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        dw[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        db[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // Row loop, software-pipelined for NV <= 4: the loads of the wave's next row (x, dy, the residual gradient, mean, rstd) are
    // issued before the current row's two wave reductions, so a row costs one memory latency instead of two in series.
    constexpr bool PIPE = NV <= 4;
    constexpr bool HOIST = NV <= 8;      // residual-gradient row loaded together with x / dy (registers permitting)
    f32x4 xv[PIPE ? 2 : 1][NV], gv[PIPE ? 2 : 1][NV], av[PIPE ? 2 : 1][HOIST ? NV : 1];
    float mr[2][2];
    auto load_row = [&](long row, int buf) {
        const float* xr = x + map_row(xm, row);
        const float* gr = dy + row * D;
        const float* ar = (HOIST && add_src) ? add_src + map_row(dxm, row) : nullptr;
        mr[buf][0] = mean_in[row];
        mr[buf][1] = rstd_in[row];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + i * 64;
            if (c < nvec) {
                xv[buf][i] = *reinterpret_cast<const f32x4*>(xr + c * 4);
                gv[buf][i] = *reinterpret_cast<const f32x4*>(gr + c * 4);
                if constexpr (HOIST) av[buf][i] = ar ? *reinterpret_cast<const f32x4*>(ar + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    auto do_row = [&](long row, int buf) {
        const float mean = mr[buf][0], rstd = mr[buf][1];
        f32x4 xh[NV];
        unsigned pos[NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + i * 64;
            if (c < nvec) {
                pos[i] = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    pos[i] |= (xv[buf][i][j] > 0.f ? 1u : 0u) << j;
                    xh[i][j] = (xv[buf][i][j] - mean) * rstd;
                    const float gw = gv[buf][i][j] * wv[i][j];
                    s1 += gw;
                    s2 += gw * xh[i][j];
                    dw[i][j] += gv[buf][i][j] * xh[i][j];
                    db[i][j] += gv[buf][i][j];
                }
            }
        }
        s1 = wave_sum(s1) / (float)D;
        s2 = wave_sum(s2) / (float)D;
        const long doff = map_row(dxm, row);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + i * 64;
            if (c < nvec) {
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = rstd * (gv[buf][i][j] * wv[i][j] - s1 - xh[i][j] * s2);
                if (relu_mask) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = ((pos[i] >> j) & 1u) ? o[j] : 0.f;
                }
                if constexpr (HOIST) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] += av[buf][i][j];
                } else if (add_src) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(add_src + doff + c * 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] += a[j];
                }
                *reinterpret_cast<f32x4*>(dx + doff + c * 4) = o;
                if (dx16) store_bf16x4(dx16 + row * D + c * 4, o);      // dense bf16 copy: the A operand of the next input-gradient GEMM
            }
        }
    };
    if constexpr (PIPE) {
        long row = r0 + wave;
        if (row < r1) load_row(row, 0);
        for (; row < r1; row += 8) {          // two rows per trip so that the buffer index is a compile-time constant
            if (row + 4 < r1) load_row(row + 4, 1);
            do_row(row, 0);
            if (row + 4 < r1) {
                if (row + 8 < r1) load_row(row + 8, 0);
                do_row(row + 4, 1);
            }
        }
    } else {
        for (long row = r0 + wave; row < r1; row += 4) {
            load_row(row, 0);
            do_row(row, 0);
        }
    }
    if (!part) return;
    // cross-wave fold: waves 1..3 park their sums in LDS, wave 0 adds them in fixed order
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + i * 64;
            if (c < nvec) {
                *reinterpret_cast<f32x4*>(&red[((wave - 1) * 2 + 0) * D + c * 4]) = dw[i];
                *reinterpret_cast<f32x4*>(&red[((wave - 1) * 2 + 1) * D + c * 4]) = db[i];
            }
        }
    }
    __syncthreads();
    if (wave == 0) {
        float* pw = part + (long)blockIdx.x * 2 * D;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + i * 64;
            if (c < nvec) {
                f32x4 a = dw[i], bb = db[i];
                for (int k = 0; k < 3; ++k) {
                    const f32x4 t0 = *reinterpret_cast<const f32x4*>(&red[(k * 2 + 0) * D + c * 4]);
                    const f32x4 t1 = *reinterpret_cast<const f32x4*>(&red[(k * 2 + 1) * D + c * 4]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { a[j] += t0[j]; bb[j] += t1[j]; }
                }
                *reinterpret_cast<f32x4*>(pw + c * 4) = a;
                *reinterpret_cast<f32x4*>(pw + D + c * 4) = bb;
            }
        }
    }
}

// D = 384 backward, two rows per wave (the forward's trick, ln_fwd_pair96_kernel): a row is 96 float4 = 1.5 per lane, so ln_bwd_kernel<2>
// leaves a quarter of its load slots empty; a PAIR of rows is 192 float4 = 3 per lane and tensor, every lane busy, and the three tensors'
// loads of a pair (x, dy, residual gradient: 9 float4 per lane) are all in flight before the first reduction.  Lane l holds vectors
// c = l, 64 + l, 128 + l of the concatenated pair; c < 96 belongs to the first row.  Four segmented wave sums per pair (two per row, as
// before).  A lane's accumulator slot i always covers column col[i]; every column is covered by two (lane, slot) pairs - one from each
// row of the pair - which the block-level fold adds in a fixed order.
__global__ __launch_bounds__(256) void ln_bwd_pair96_kernel(const float* __restrict__ x, RowMap xm, const float* __restrict__ dy,
                                                            const float* __restrict__ w, const float* __restrict__ mean_in,
                                                            const float* __restrict__ rstd_in, float* __restrict__ dx,
                                                            const float* __restrict__ add_src, float* __restrict__ part, long rows,
                                                            long rows_per_block, int relu_mask, __bf16* __restrict__ dx16) {
    constexpr int NVEC = 96, D = 384;
    __shared__ __attribute__((aligned(16))) float red[4][2][D];      // [wave][dw | db][column]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = min(rows, r0 + rows_per_block);
    int col[3];
    bool second[3];
    f32x4 wv[3], dw[3], db[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = lane + 64 * i;
        second[i] = c >= NVEC;
        col[i] = second[i] ? c - NVEC : c;
        wv[i] = *reinterpret_cast<const f32x4*>(w + col[i] * 4);
        dw[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        db[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (long ra = r0 + 2 * wave; ra < r1; ra += 8) {
        const bool has1 = ra + 1 < r1;
        const long rb = has1 ? ra + 1 : ra;
        const long oa = map_row(xm, ra), ob = map_row(xm, rb);
        f32x4 xv[3], gv[3], av[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            xv[i] = *reinterpret_cast<const f32x4*>(x + (second[i] ? ob : oa) + col[i] * 4);
            gv[i] = *reinterpret_cast<const f32x4*>(dy + (second[i] ? rb : ra) * D + col[i] * 4);
            av[i] = add_src ? *reinterpret_cast<const f32x4*>(add_src + (second[i] ? ob : oa) + col[i] * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const float mean_a = mean_in[ra], rstd_a = rstd_in[ra], mean_b = mean_in[rb], rstd_b = rstd_in[rb];
        f32x4 xh[3];
        float s1a = 0.f, s2a = 0.f, s1b = 0.f, s2b = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float mean = second[i] ? mean_b : mean_a, rstd = second[i] ? rstd_b : rstd_a;
            const bool live = !second[i] || has1;           // the odd last row of a block: its "second" half repeats row a and is dropped
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                xh[i][j] = (xv[i][j] - mean) * rstd;
                const float gw = gv[i][j] * wv[i][j];
                t1 += gw;
                t2 += gw * xh[i][j];
                if (live) {
                    dw[i][j] += gv[i][j] * xh[i][j];
                    db[i][j] += gv[i][j];
                }
            }
            if (second[i]) { s1b += t1; s2b += t2; } else { s1a += t1; s2a += t2; }
        }
        s1a = wave_sum(s1a) / (float)D;
        s2a = wave_sum(s2a) / (float)D;
        s1b = wave_sum(s1b) / (float)D;
        s2b = wave_sum(s2b) / (float)D;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (second[i] && !has1) continue;
            const float rstd = second[i] ? rstd_b : rstd_a, s1 = second[i] ? s1b : s1a, s2 = second[i] ? s2b : s2a;
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = rstd * (gv[i][j] * wv[i][j] - s1 - xh[i][j] * s2);
                if (relu_mask && !(xv[i][j] > 0.f)) o[j] = 0.f;
                o[j] += av[i][j];
            }
            const long off = (second[i] ? ob : oa) + col[i] * 4;
            *reinterpret_cast<f32x4*>(dx + off) = o;
            if (dx16) store_bf16x4(dx16 + (second[i] ? rb : ra) * D + col[i] * 4, o);
        }
    }
    if (!part) return;
    // fold: per wave, first the (lane, slot) pairs that came from first rows, then - after a barrier - the pairs from second rows are added
    // to the same columns; then the four waves' sums in wave order (deterministic)
#pragma unroll
    for (int i = 0; i < 3; ++i)
        if (!second[i]) {
            *reinterpret_cast<f32x4*>(&red[wave][0][col[i] * 4]) = dw[i];
            *reinterpret_cast<f32x4*>(&red[wave][1][col[i] * 4]) = db[i];
        }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 3; ++i)
        if (second[i]) {
            f32x4 a = *reinterpret_cast<const f32x4*>(&red[wave][0][col[i] * 4]);
            f32x4 b = *reinterpret_cast<const f32x4*>(&red[wave][1][col[i] * 4]);
#pragma unroll
            for (int j = 0; j < 4; ++j) { a[j] += dw[i][j]; b[j] += db[i][j]; }
            *reinterpret_cast<f32x4*>(&red[wave][0][col[i] * 4]) = a;
            *reinterpret_cast<f32x4*>(&red[wave][1][col[i] * 4]) = b;
        }
    __syncthreads();
    float* pw = part + (long)blockIdx.x * 2 * D;
    for (int c = threadIdx.x; c < 2 * D; c += 256) {
        const int which = c >= D, cc = which ? c - D : c;
        pw[c] = (red[0][which][cc] + red[1][which][cc]) + (red[2][which][cc] + red[3][which][cc]);
    }
}

// fold the per-block partials: one workgroup per 64 columns of [dweight | dbias]; the 4 waves split the partial
// index, LDS combines them in a fixed order (deterministic).
// 1024 threads = 16 columns x 64 partial-lanes: each thread adds every 64th partial of its column (few, independent loads), LDS
// then combines the 64 lane sums of a column in a fixed order.  grid = ceil(2D / 16): 48 workgroups at D = 384 instead of 12.
constexpr int FOLD_COLS = 16, FOLD_LANES = 64;
__global__ __launch_bounds__(FOLD_COLS * FOLD_LANES) void ln_bwd_fold_kernel(const float* __restrict__ part, int nblocks, int D,
                                                                             float* __restrict__ dw, float* __restrict__ db,
                                                                             int accumulate) {
    __shared__ float red[FOLD_LANES][FOLD_COLS + 1];
    const int col = threadIdx.x & (FOLD_COLS - 1), kl = threadIdx.x / FOLD_COLS;
    const int c = blockIdx.x * FOLD_COLS + col;
    float s = 0.f;
    if (c < 2 * D)
        for (int k = kl; k < nblocks; k += FOLD_LANES) s += part[(long)k * 2 * D + c];
    red[kl][col] = s;
    __syncthreads();
    if (kl == 0 && c < 2 * D) {
        float t = 0.f;
#pragma unroll 8
        for (int k = 0; k < FOLD_LANES; ++k) t += red[k][col];
        float* o = c < D ? dw + c : db + (c - D);
        *o = accumulate ? *o + t : t;
    }
}

// ---- generic scalar path for feature widths that are not a multiple of 4 (T2T stage 1: 3*7*7 = 147) ----
__global__ __launch_bounds__(256) void ln_fwd_scalar_kernel(const float* __restrict__ x, RowMap xm, const float* __restrict__ w,
                                                            const float* __restrict__ b, float* __restrict__ y,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                            long rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + map_row(xm, row);
    float s = 0.f;
    for (int c = lane; c < D; c += 64) s += xr[c];
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
    for (int c = lane; c < D; c += 64) { const float d = xr[c] - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
    for (int c = lane; c < D; c += 64) y[row * D + c] = (xr[c] - mean) * rstd * w[c] + b[c];
    if (lane == 0) {
        if (mean_out) mean_out[row] = mean;
        if (rstd_out) rstd_out[row] = rstd;
    }
}

// one workgroup per chunk of rows; dweight / dbias partials per block via per-thread column ownership (column c is
// always handled by lane c % 64 of every wave, so the four waves' sums are combined through LDS)
__global__ __launch_bounds__(256) void ln_bwd_scalar_kernel(const float* __restrict__ x, RowMap xm, const float* __restrict__ dy,
                                                            const float* __restrict__ w, const float* __restrict__ mean_in,
                                                            const float* __restrict__ rstd_in, float* __restrict__ dx,
                                                            const float* __restrict__ add_src, float* __restrict__ part, long rows,
                                                            int D, long rows_per_block, int relu_mask) {
    extern __shared__ __attribute__((aligned(16))) float red[];  // [4][2][D]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    for (int c = lane; c < 2 * D; c += 64) red[wave * 2 * D + c] = 0.f;
    for (long row = r0 + wave; row < r1; row += 4) {
        const long off = map_row(xm, row);
        const float* xr = x + off;
        const float* gr = dy + row * D;
        const float mean = mean_in[row], rstd = rstd_in[row];
        float s1 = 0.f, s2 = 0.f;
        for (int c = lane; c < D; c += 64) {
            const float xh = (xr[c] - mean) * rstd, gw = gr[c] * w[c];
            s1 += gw;
            s2 += gw * xh;
            red[(wave * 2 + 0) * D + c] += gr[c] * xh;
            red[(wave * 2 + 1) * D + c] += gr[c];
        }
        s1 = wave_sum(s1) / (float)D;
        s2 = wave_sum(s2) / (float)D;
        for (int c = lane; c < D; c += 64) {
            const float xv = xr[c], xh = (xv - mean) * rstd;
            float o = rstd * (gr[c] * w[c] - s1 - xh * s2);
            if (relu_mask && !(xv > 0.f)) o = 0.f;
            if (add_src) o += add_src[off + c];
            dx[off + c] = o;
        }
    }
    __syncthreads();
    if (part)
        for (int c = threadIdx.x; c < 2 * D; c += 256)
            part[(long)blockIdx.x * 2 * D + c] = (red[c] + red[2 * D + c]) + (red[4 * D + c] + red[6 * D + c]);
}

inline int pick_nv(int D) {
    const int nvec = D / 4;
    if (nvec <= 64) return 1;
    if (nvec <= 128) return 2;
    if (nvec <= 256) return 4;
    if (nvec <= 512) return 8;
    return 16;
}
inline int bwd_blocks(long rows) {
    // one wave walks its rows one after the other (load -> two wave reductions -> store), so the kernel is latency-bound unless
    // several workgroups share a CU: up to 3 per CU (measured: 256 blocks 48.8 us, 768 blocks see DESIGN.md section 7)
    static const long cap = [] { const char* e = getenv("D2S_LN_BWD_BLOCKS"); return e ? atol(e) : 768L; }();
    long nb = (rows + 31) / 32;
    if (nb > cap) nb = cap;
    if (nb < 1) nb = 1;
    return (int)nb;
}

}  // namespace

extern "C" {

// Row addressing for x (and dx in backward): element offset of logical row r is
//   (r / rows_per_group) * group_stride + offset + (r % rows_per_group) * row_stride.
// Contiguous [rows, D]: rows_per_group = rows, group_stride = 0, row_stride = D, offset = 0.
static int layernorm_fwd_impl(const float* x, long rows_per_group, long group_stride, long row_stride, long offset, const float* w,
                              const float* b, float* y, __bf16* y16, float* mean, float* rstd, long rows, int D, float eps, hipStream_t stream) {
    if (!x || !w || !b || (!y && !y16) || rows <= 0 || D <= 0 || D > 4096 || rows_per_group <= 0) return D2S_ERR_ARG;
    RowMap m{rows_per_group, group_stride, row_stride, offset};
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    if ((D & 3) || ((group_stride | row_stride | offset) & 3)) {
        if (y16 || !y) return D2S_ERR_ARG;      // the bf16 copy exists for the vector kernels only (D % 4 == 0)
        hipLaunchKernelGGL(ln_fwd_scalar_kernel, grid, block, 0, stream, x, m, w, b, y, mean, rstd, rows, D, eps);
        return d2s_check_launch();
    }
    static const int pair_env = [] { const char* e = getenv("D2S_LN_PAIR"); return e ? atoi(e) : 1; }();
    if (D == 384 && pair_env) {
        hipLaunchKernelGGL(ln_fwd_pair96_kernel, dim3((unsigned)((rows + 7) / 8)), block, 0, stream, x, m, w, b, y, y16, mean, rstd, rows, eps);
        return d2s_check_launch();
    }
    switch (pick_nv(D)) {
        case 1: hipLaunchKernelGGL(ln_fwd_kernel<1>, grid, block, 0, stream, x, m, w, b, y, y16, mean, rstd, rows, D, eps); break;
        case 2: hipLaunchKernelGGL(ln_fwd_kernel<2>, grid, block, 0, stream, x, m, w, b, y, y16, mean, rstd, rows, D, eps); break;
        case 4: hipLaunchKernelGGL(ln_fwd_kernel<4>, grid, block, 0, stream, x, m, w, b, y, y16, mean, rstd, rows, D, eps); break;
        case 8: hipLaunchKernelGGL(ln_fwd_kernel<8>, grid, block, 0, stream, x, m, w, b, y, y16, mean, rstd, rows, D, eps); break;
        default: hipLaunchKernelGGL(ln_fwd_kernel<16>, grid, block, 0, stream, x, m, w, b, y, y16, mean, rstd, rows, D, eps); break;
    }
    return d2s_check_launch();
}

int d2s_layernorm_fwd(const float* x, long rows_per_group, long group_stride, long row_stride, long offset, const float* w,
                      const float* b, float* y, float* mean, float* rstd, long rows, int D, float eps, hipStream_t stream) {
    if (!y) return D2S_ERR_ARG;
    return layernorm_fwd_impl(x, rows_per_group, group_stride, row_stride, offset, w, b, y, nullptr, mean, rstd, rows, D, eps, stream);
}

// The same LayerNorm with a dense [rows, D] bf16 copy of the output for a following bf16-mode GEMM (its a_bf16); y may be NULL when only
// the bf16 form is consumed (forward-only passes).  D % 4 == 0.
int d2s_layernorm_fwd_bf16out(const float* x, long rows_per_group, long group_stride, long row_stride, long offset, const float* w,
                              const float* b, float* y, void* y_bf16, float* mean, float* rstd, long rows, int D, float eps,
                              hipStream_t stream) {
    if (!y_bf16) return D2S_ERR_ARG;
    return layernorm_fwd_impl(x, rows_per_group, group_stride, row_stride, offset, w, b, y, static_cast<__bf16*>(y_bf16), mean, rstd, rows, D,
                              eps, stream);
}

size_t d2s_layernorm_bwd_workspace_bytes(long rows, int D) { return (size_t)bwd_blocks(rows) * 2 * D * sizeof(float); }

// dx[map(r)] = (add_src ? add_src[map(r)] : 0) + mask * dLN/dx ; dweight / dbias (+)= column sums (skipped if dweight
// null).  relu_mask != 0: mask = (x > 0), i.e. the backward of a ReLU whose OUTPUT is this LayerNorm's input
// (the predictor's Linear->ReLU->LayerNorm chain, dynamic_vit.py:515-528), folded into the same pass.
// x and dx/add_src share one RowMap; dy, mean, rstd are contiguous per logical row.
static int layernorm_bwd_impl(const float* x, long rows_per_group, long group_stride, long row_stride, long offset, const float* dy,
                              const float* w, const float* mean, const float* rstd, float* dx, __bf16* dx16, const float* add_src, float* dweight,
                              float* dbias, int accumulate_wb, int relu_mask, long rows, int D, void* workspace,
                              size_t workspace_bytes, hipStream_t stream) {
    if (!x || !dy || !w || !mean || !rstd || !dx || rows <= 0 || D <= 0 || D > 4096) return D2S_ERR_ARG;
    const bool scalar = (D & 3) || ((group_stride | row_stride | offset) & 3);
    const int nb = bwd_blocks(rows);
    float* part = nullptr;
    if (dweight) {
        if (!dbias) return D2S_ERR_ARG;
        if (!workspace || workspace_bytes < (size_t)nb * 2 * D * sizeof(float)) return D2S_ERR_WORKSPACE;
        part = static_cast<float*>(workspace);
    }
    RowMap m{rows_per_group, group_stride, row_stride, offset};
    const long rpb = (rows + nb - 1) / nb;
    const int nblocks = (int)((rows + rpb - 1) / rpb);
    dim3 grid(nblocks), block(256);
    const size_t sh = (size_t)3 * 2 * D * sizeof(float);
    if (scalar) {
        if (dx16) return D2S_ERR_ARG;      // the bf16 copy exists for the vector kernels only (D % 4 == 0)
        hipLaunchKernelGGL(ln_bwd_scalar_kernel, grid, block, (size_t)4 * 2 * D * sizeof(float), stream, x, m, dy, w, mean, rstd, dx,
                           add_src, part, rows, D, rpb, relu_mask);
        if (dweight)
            hipLaunchKernelGGL(ln_bwd_fold_kernel, dim3((2 * D + FOLD_COLS - 1) / FOLD_COLS), dim3(FOLD_COLS * FOLD_LANES), 0, stream, part, nblocks, D, dweight, dbias,
                               accumulate_wb);
        return d2s_check_launch();
    }
    static const int bwd_pair_env = [] { const char* e = getenv("D2S_LN_BWD_PAIR"); return e ? atoi(e) : 1; }();
    if (D == 384 && bwd_pair_env) {      // two rows per wave (D2S_LN_BWD_PAIR=0: the generic kernel)
        hipLaunchKernelGGL(ln_bwd_pair96_kernel, grid, block, 0, stream, x, m, dy, w, mean, rstd, dx, add_src, part, rows, rpb, relu_mask, dx16);
        if (dweight)
            hipLaunchKernelGGL(ln_bwd_fold_kernel, dim3((2 * D + FOLD_COLS - 1) / FOLD_COLS), dim3(FOLD_COLS * FOLD_LANES), 0, stream, part, nblocks, D, dweight, dbias,
                               accumulate_wb);
        return d2s_check_launch();
    }
#define D2S_LN_BWD(NV) hipLaunchKernelGGL(ln_bwd_kernel<NV>, grid, block, sh, stream, x, m, dy, w, mean, rstd, dx, m, add_src, part, rows, D, rpb, relu_mask, dx16)
    switch (pick_nv(D)) {
        case 1: D2S_LN_BWD(1); break;
        case 2: D2S_LN_BWD(2); break;
        case 4: D2S_LN_BWD(4); break;
        case 8: D2S_LN_BWD(8); break;
        default: D2S_LN_BWD(16); break;
    }
#undef D2S_LN_BWD
    if (dweight)
        hipLaunchKernelGGL(ln_bwd_fold_kernel, dim3((2 * D + FOLD_COLS - 1) / FOLD_COLS), dim3(FOLD_COLS * FOLD_LANES), 0, stream, part, nblocks, D, dweight, dbias,
                           accumulate_wb);
    return d2s_check_launch();
}

int d2s_layernorm_bwd(const float* x, long rows_per_group, long group_stride, long row_stride, long offset, const float* dy,
                      const float* w, const float* mean, const float* rstd, float* dx, const float* add_src, float* dweight,
                      float* dbias, int accumulate_wb, int relu_mask, long rows, int D, void* workspace,
                      size_t workspace_bytes, hipStream_t stream) {
    return layernorm_bwd_impl(x, rows_per_group, group_stride, row_stride, offset, dy, w, mean, rstd, dx, nullptr, add_src, dweight, dbias,
                              accumulate_wb, relu_mask, rows, D, workspace, workspace_bytes, stream);
}

// The same backward with a dense [rows, D] bf16 copy of dx (after the residual add): the a_bf16 of the input-gradient GEMM that consumes
// it in the bf16 mode.  D % 4 == 0.
int d2s_layernorm_bwd_bf16out(const float* x, long rows_per_group, long group_stride, long row_stride, long offset, const float* dy,
                              const float* w, const float* mean, const float* rstd, float* dx, void* dx_bf16, const float* add_src,
                              float* dweight, float* dbias, int accumulate_wb, int relu_mask, long rows, int D, void* workspace,
                              size_t workspace_bytes, hipStream_t stream) {
    if (!dx_bf16) return D2S_ERR_ARG;
    return layernorm_bwd_impl(x, rows_per_group, group_stride, row_stride, offset, dy, w, mean, rstd, dx, static_cast<__bf16*>(dx_bf16), add_src,
                              dweight, dbias, accumulate_wb, relu_mask, rows, D, workspace, workspace_bytes, stream);
}

}  // extern "C"
