// dsn_train.hip - parameter gradients of Renderer.render (trainer.py:70-81 loss.backward()).
//
// What autograd does in the reference, restated analytically.  With per-sample cotangents dL/dsigma, dL/dessence,
// dL/dn_w coming back from compositing (utils/nerf_net_utils.py:5-56), the colour product and the lighting MLP
// (model/spacenet.py:174-188, :254-265), the parameters of the canonical field receive
//
//   dW_l = sum_n  (m_l * ahat_l)[n] (x) h_{l-1}[n]   +   (m_l * a_l)[n] (x) hdot_{l-1}[n]
//
//   h_l     forward activations (model/spacenet.py:93-148), m_l their ReLU patterns
//   a_l     adjoint of sigma (the reverse pass that also yields g = d sigma/dx, model/spacenet.py:301-311)
//   ahat_l  adjoint of  dL/dsigma * sigma + dL/dessence . essence
//   hdot_l  forward TANGENT along u = dL/dg:  u . grad_x sigma is the directional derivative of sigma along u, a
//           network that is linear in every W_l with the same patterns m_l, so its weight gradient is the outer
//           product of the sigma-adjoint from above and the tangent from below.  This is the double-backward term
//           through  grad sigma -> n_w -> lighting  (model/spacenet.py:251-265) without a second autograd graph.
//   u       = J^T dL/dn_w, J = d n_w / d g: projection onto the canonical nearest face and re-embedding on the posed
//           face are affine in the point (utils/geo_utils.py:96-113,138-156,181-200), then F.normalize.
//
// This round's implementation: activations of one training batch resident in HBM (23 KB per sample - 12 GB for the
// 8192 x 64 batch of BASELINE configs[2], sized for 288 GB).  The forward pass, the sigma reverse pass and the tangent
// pass and the adjoint pass are fused split-fp16 kernels of dsn_field16.hip (k_field16<train> stores h_l, a_l and the relu
// records as it goes; k_tangent16 stores hdot_l; k_adjoint16 stores ahat_l below its seed); the small lighting / rgb-head
// products run on k_t_lin (fp32 MFMA, fused bias / mask / seed epilogues; no library GEMM is left); the weight-gradient products, which contract over the half-million samples of the
// batch, run on the hand-written exact-fp32 MFMA kernel k_t_wgrad below; everything else is element-wise kernels.
#include "dsn_common.h"
#include "dsn_kernels.h"


namespace {

enum { P_EMB = 0, P_S1_0W, P_S1_0B, P_S1_2W, P_S1_2B, P_S1_4W, P_S1_4B, P_S1_6W, P_S1_6B, P_S2_0W, P_S2_0B, P_S2_2W,
       P_S2_2B, P_S2_4W, P_S2_4B, P_DEN_W, P_DEN_B, P_RGB1_W, P_RGB1_B, P_RGB3_W, P_RGB3_B, P_L0_W, P_L0_B, P_L2_W,
       P_L2_B, P_L4_W, P_L4_B, P_PM0_W, P_PM0_B, P_PM2_W, P_PM2_B, P_PM4_W, P_PM4_B };

const int kParamCount[33] = {500 * 8, 256 * 87, 256, 256 * 256, 256, 256 * 256, 256, 256 * 256, 256, 256 * 319, 256,
                             256 * 256, 256, 256 * 256, 256, 256, 1, 128 * 256, 128, 3 * 128, 3, 128 * 9, 128,
                             128 * 128, 128, 128, 1, 64 * 92, 64, 64 * 64, 64, 16 * 64, 16};

// trunk layer l: weight / bias parameter index, input width (leading dimension of W), column of the h-part
const int kTrunkW[7] = {P_S1_0W, P_S1_2W, P_S1_4W, P_S1_6W, P_S2_0W, P_S2_2W, P_S2_4W};
const int kTrunkB[7] = {P_S1_0B, P_S1_2B, P_S1_4B, P_S1_6B, P_S2_0B, P_S2_2B, P_S2_4B};
const int kTrunkLd[7] = {87, 256, 256, 256, 319, 256, 256};
#define PE_LD 64          // positional encoding rows: 63 values + one zero pad
#define PE_K 63
#define W0_PE_COL 8       // stage1.0 input = [code 8 | pe 63 | pose 16]  (model/spacenet.py:125-131)
#define W0_POSE_COL 71
#define W4_PE_COL 256     // stage2.0 input = [h 256 | pe 63]              (model/spacenet.py:133-135)

// ------------------------------------------------------------------------------------------------------------
// element-wise kernels
// ------------------------------------------------------------------------------------------------------------
#define T_THREADS 256
inline dim3 grid_for(int64_t n) { return dim3((unsigned)((n + T_THREADS - 1) / T_THREADS)); }

// Row subsets (round 3).  A sample whose cotangents are all zero contributes exactly nothing to any parameter gradient (every pass
// below is linear in them), and a sample whose alpha is exactly 0 - transparent with noise <= 0, or relu(sigma + noise) = 0 - has
// zero cotangents (utils/nerf_net_utils.py:24-39: w = alpha T; d alpha / d sigma carries relu'), so the passes run on the LISTED rows
// only: `list` = ascending sample indices (NULL: all rows), `cnt` = their number on the device (NULL: N).  Arrays stay indexed by
// sample; rows that are not listed are never read or written.  Launches are sized for N, kernels stop at the device count.
struct Rows { const int32_t* list; const int32_t* cnt; };
__device__ __forceinline__ int64_t rows_n(const Rows& r, int64_t N) { return r.cnt ? (int64_t)(*r.cnt) : N; }
__device__ __forceinline__ int64_t rows_at(const Rows& r, int64_t k) { return r.list ? (int64_t)r.list[k] : k; }
// the same for a WAVE-UNIFORM k, read through the constant address space: only for loads from there does hipcc pick a scalar load
// (s_load_dword, waited for on lgkmcnt) - a plain global pointer gives a vector load whose s_waitcnt vmcnt(0) sits between the row
// number and every load that depends on it, draining the loads in flight.  The lists are not written while their readers run.
__device__ __forceinline__ int64_t rows_at_u(const Rows& r, int64_t k) {
    typedef const __attribute__((address_space(4))) int32_t* cptr;
    return r.list ? (int64_t)*reinterpret_cast<cptr>((uintptr_t)(r.list + k)) : k;
}
// rows a workgroup of a row-chunked kernel takes: an even share of the listed rows, in multiples of `mult`, at least `least`
__device__ __forceinline__ int64_t rows_share_of(int64_t NL, int64_t groups, int mult, int least) {
    int64_t rows = (NL + groups - 1) / groups;
    if (rows < least) rows = least;
    return (rows + mult - 1) / mult * mult;
}
__device__ __forceinline__ int64_t rows_share(int64_t NL, int mult, int least) { return rows_share_of(NL, (int64_t)gridDim.x, mult, least); }

// ascending list of the rows with flag != 0: per-block counts, single-block scan, fill (deterministic; N / 256 <= 4096 blocks)
__global__ void __launch_bounds__(T_THREADS) k_t_rows_count(const uint8_t* __restrict__ flag, int64_t N, int32_t* __restrict__ bcnt) {
    const int64_t n = (int64_t)blockIdx.x * T_THREADS + threadIdx.x;
    const bool on = n < N && flag[n] != 0;
    __shared__ int s_c[T_THREADS / 64];
    const unsigned long long m = __ballot(on);
    if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) bcnt[blockIdx.x] = s_c[0] + s_c[1] + s_c[2] + s_c[3];
}
__global__ void __launch_bounds__(1024) k_t_rows_scan(int32_t* __restrict__ bcnt, int nb, int32_t* __restrict__ total) {
    __shared__ int s_w[16];
    __shared__ int s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < nb ? bcnt[i] : 0;
        int inc = v;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int x = __shfl_up(inc, o); if (lane >= o) inc += x; }
        if (lane == 63) s_w[wave] = inc;
        __syncthreads();
        int wp = 0, tot = 0;
        for (int k = 0; k < 16; ++k) { const int x = s_w[k]; if (k < wave) wp += x; tot += x; }
        const int carry = s_carry;
        if (i < nb) bcnt[i] = carry + wp + inc - v;      // exclusive prefix
        __syncthreads();
        if (threadIdx.x == 0) s_carry = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = s_carry;
}
__global__ void __launch_bounds__(T_THREADS) k_t_rows_fill(const uint8_t* __restrict__ flag, int64_t N, const int32_t* __restrict__ boff,
                                                            int32_t* __restrict__ list) {
    const int64_t n = (int64_t)blockIdx.x * T_THREADS + threadIdx.x;
    const bool on = n < N && flag[n] != 0;
    __shared__ int s_c[T_THREADS / 64];
    const unsigned long long m = __ballot(on);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_c[wave] = __popcll(m);
    __syncthreads();
    if (on) {
        int off = boff[blockIdx.x] + __popcll(m & ((1ull << lane) - 1ull));
        for (int k = 0; k < wave; ++k) off += s_c[k];
        list[off] = (int32_t)n;
    }
}
// module mode (explicit cotangents): rows with a non-zero cotangent
__global__ void __launch_bounds__(T_THREADS) k_t_flag_cotangent(const float* __restrict__ d_col, const float* __restrict__ d_sig, int64_t N,
                                                                 uint8_t* __restrict__ flag) {
    const int64_t n = (int64_t)blockIdx.x * T_THREADS + threadIdx.x;
    if (n >= N) return;
    flag[n] = (d_sig[n] != 0.0f || d_col[3 * n] != 0.0f || d_col[3 * n + 1] != 0.0f || d_col[3 * n + 2] != 0.0f) ? 1 : 0;
}
// live rows of the FORWARD: everything except transparent samples whose noise is <= 0 (alpha = 0 exactly: neither their colour nor
// their density reaches an output or receives a gradient, can_render.py:115-120 + utils/nerf_net_utils.py:30-36)
__global__ void __launch_bounds__(T_THREADS) k_t_flag_forward(const uint8_t* __restrict__ transparent, const float* __restrict__ noise,
                                                               int64_t N, uint8_t* __restrict__ flag) {
    const int64_t n = (int64_t)blockIdx.x * T_THREADS + threadIdx.x;
    if (n >= N) return;
    flag[n] = (transparent[n] == 0 || (noise && noise[n] > 0.0f)) ? 1 : 0;
}

// model/dimension_kernel.py:34-35,56-75: [x, sin(2^j x), cos(2^j x)]_{j<10}; column 63 is a zero pad
__global__ void __launch_bounds__(T_THREADS) k_t_pe(const float* __restrict__ x_c, int64_t N, float* __restrict__ pe, Rows rw) {
    const int64_t tl = (int64_t)blockIdx.x * T_THREADS + threadIdx.x;
    if (tl >= rows_n(rw, N) * PE_LD) return;
    const int64_t n = rows_at(rw, tl >> 6);
    const int c = (int)(tl & 63);
    const int64_t t = n * PE_LD + c;
    float v = 0.0f;
    if (c < 3) v = x_c[3 * n + c];
    else if (c < PE_K) {
        const int j = (c - 3) / 6, r = (c - 3) % 6, a = r % 3;
        const float arg = x_c[3 * n + a] * (float)(1 << j);
        v = r < 3 ? sinf(arg) : cosf(arg);
    }
    pe[t] = v;
}

// tangent of the encoding along u: [u, 2^j cos(2^j x) u, -2^j sin(2^j x) u]
// (gmax: batch-wide max |tpe| as float bits, for the split-fp16 weight-gradient product; zeroed by the caller.  Grid-stride
// loop: one look-before-atomicMax per BLOCK - half a million waves reading the same word serialise on its L2 channel)
// pe (optional, round 5): the encoding itself rides along (same sinf / cosf of the same argument as k_t_pe: one launch and one sweep less)
__global__ void __launch_bounds__(T_THREADS) k_t_pe_tangent(const float* __restrict__ x_c, const float* __restrict__ u,
                                                             int64_t N, float* __restrict__ tpe, unsigned* __restrict__ gmax, Rows rw,
                                                             float* __restrict__ pe) {
    __shared__ float s_m[T_THREADS / 64];
    float m = 0.0f;
    const int64_t total = rows_n(rw, N) * PE_LD, stride = (int64_t)gridDim.x * T_THREADS;
    for (int64_t tl = (int64_t)blockIdx.x * T_THREADS + threadIdx.x; tl < total; tl += stride) {
        const int64_t n = rows_at(rw, tl >> 6);
        const int c = (int)(tl & 63);
        const int64_t t = n * PE_LD + c;
        float v = 0.0f, e = 0.0f;
        if (c < 3) { v = u[3 * n + c]; if (pe) e = x_c[3 * n + c]; }
        else if (c < PE_K) {
            const int j = (c - 3) / 6, r = (c - 3) % 6, a = r % 3;
            const float f = (float)(1 << j);
            const float arg = x_c[3 * n + a] * f;
            const float sn = (r >= 3 || pe) ? sinf(arg) : 0.0f, cs = (r < 3 || pe) ? cosf(arg) : 0.0f;
            v = (r < 3 ? cs : -sn) * f * u[3 * n + a];
            e = r < 3 ? sn : cs;
        }
        tpe[t] = v;
        if (pe) pe[t] = e;
        const float av = fabsf(v);
        if (av < 3.0e38f) m = fmaxf(m, av);      // non-finite values do not set the scale
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < T_THREADS / 64; ++k) m = fmaxf(m, s_m[k]);
        if (m > 0.0f && __float_as_uint(m) > __atomic_load_n(gmax, __ATOMIC_RELAXED)) atomicMax(gmax, __float_as_uint(m));
    }
}

// g = J_pe^T dpe  (model/spacenet.py:301-311 through the encoding)
__global__ void __launch_bounds__(T_THREADS) k_t_pe_reverse(const float* __restrict__ x_c, const float* __restrict__ dpe,
                                                             int64_t N, float* __restrict__ g) {
    const int64_t t = (int64_t)blockIdx.x * T_THREADS + threadIdx.x;
    if (t >= N * 3) return;
    const int64_t n = t / 3;
    const int a = (int)(t % 3);
    const float x = x_c[t];
    const float* d = dpe + n * PE_LD;
    float acc = d[a];
    for (int j = 0; j < 10; ++j) {
        const float f = (float)(1 << j);
        const float arg = x * f;
        acc += f * (cosf(arg) * d[3 + 6 * j + a] - sinf(arg) * d[6 + 6 * j + a]);
    }
    g[t] = acc;
}

// out[n,c] = (h[n,c] > 0) ? base[n,c] (optional) + scale[n] (optional, else 1) * w[c] : 0
__global__ void __launch_bounds__(T_THREADS) k_t_seed(const float* __restrict__ h, const float* __restrict__ w,
                                                       const float* __restrict__ scale, const float* base, int C,
                                                       int64_t total, float* out, Rows rw) {   // base may alias out
    const int64_t tl = 4 * ((int64_t)blockIdx.x * T_THREADS + threadIdx.x);
    if (tl >= rows_n(rw, total / C) * C) return;
    const int64_t n = rows_at(rw, tl / C);
    const int c = (int)(tl % C);
    const int64_t t = n * C + c;
    const float sc = scale ? scale[n] : 1.0f;
    const float4 ww = make_float4(w[c], w[c + 1], w[c + 2], w[c + 3]);
    float4 v = make_float4(sc * ww.x, sc * ww.y, sc * ww.z, sc * ww.w);
    if (base) {
        const float4 bb = *reinterpret_cast<const float4*>(base + t);
        v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
    }
    const float4 hh = *reinterpret_cast<const float4*>(h + t);
    v.x = hh.x > 0.0f ? v.x : 0.0f; v.y = hh.y > 0.0f ? v.y : 0.0f; v.z = hh.z > 0.0f ? v.z : 0.0f; v.w = hh.w > 0.0f ? v.w : 0.0f;
    *reinterpret_cast<float4*>(out + t) = v;
}

__device__ __forceinline__ float t_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// out[n, k] = bias[k] + h[n,:] . w[k,:]   (K <= 3 output rows; one wave per sample)
__global__ void __launch_bounds__(T_THREADS) k_t_rowdot(const float* __restrict__ h, int C, const float* __restrict__ w,
                                                         const float* __restrict__ bias, int K, int64_t N,
                                                         float* __restrict__ out, Rows rw) {
    const int lane = threadIdx.x & 63;
    const int64_t kr = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (kr >= rows_n(rw, N)) return;
    const int64_t n = rows_at(rw, kr);
    for (int k = 0; k < K; ++k) {
        float acc = 0.0f;
        for (int c = lane; c < C; c += 64) acc = fmaf(h[n * C + c], w[k * C + c], acc);
        acc = t_wave_sum(acc);
        if (lane == 0) out[n * K + k] = acc + bias[k];
    }
}

// a = (h > 0) ? a : 0 on a [N,256] matrix AND its column sums accumulated into out[256] (one pass instead of two)
__global__ void __launch_bounds__(T_THREADS) k_t_mask_colsum256(float* __restrict__ a, const float* __restrict__ h, int64_t N,
                                                                 int rows_per_block, float* __restrict__ out) {
    const int c = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * rows_per_block;
    int64_t end = base + rows_per_block;
    if (end > N) end = N;
    float acc = 0.0f;
    for (int64_t n = base; n < end; ++n) {
        const int64_t t = n * 256 + c;
        float v = a[t];
        if (!(h[t] > 0.0f)) { v = 0.0f; a[t] = 0.0f; }
        acc += v;
    }
    atomicAdd(out + c, acc);
}

// column sums of a [N,C] matrix (C <= 256), accumulated into out[C]
__global__ void __launch_bounds__(T_THREADS) k_t_colsum(const float* __restrict__ a, int C, int64_t N, int rows_per_block,
                                                         float* __restrict__ out, Rows rw) {
    __shared__ float s[T_THREADS];
    // (C is 128 or 256: a wave lies inside one row group, so the row numbers are wave-uniform - scalar loads, rows_at_u)
    const int c = threadIdx.x % C, r0 = __builtin_amdgcn_readfirstlane(threadIdx.x / C), rs = T_THREADS / C;
    const int64_t NL = rows_n(rw, N);
    const int64_t base = (int64_t)blockIdx.x * rows_per_block;
    int64_t end = base + rows_per_block;
    if (end > NL) end = NL;
    float acc = 0.0f;
    if (r0 < rs) {
        int64_t n = base + r0;
        int64_t id[8];
        // the row numbers of a step are fetched one step ahead (rows beyond the chunk: the last row's, never used)
        auto fetch = [&](int64_t n_) {
#pragma unroll
            for (int k = 0; k < 8; ++k) { const int64_t q = n_ + k * (int64_t)rs; id[k] = rows_at_u(rw, q < end ? q : end - 1); }
        };
        if (n < end) fetch(n);
        for (; n + 7 * (int64_t)rs < end; n += 8 * (int64_t)rs) {      // eight independent loads in flight per thread
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = a[id[k] * C + c];
            fetch(n + 8 * (int64_t)rs);
            acc += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
        for (; n < end; n += rs) acc += a[rows_at_u(rw, n) * C + c];
    }
    s[threadIdx.x] = acc;
    __syncthreads();
    if (r0 == 0) {
        for (int k = 1; k < rs; ++k) acc += s[k * C + c];
        atomicAdd(out + c, acc);
    }
}

// Weight gradient of a layer with 1 or 3 outputs: dW[o][c] += sum_n dY[n,o] X[n,c] (C = 128 or 256), and its bias gradient
// db[o] += sum_n dY[n,o] - a weighted column sum at the HBM rate (rocBLAS runs these K = 524 288, M <= 3 shapes at 0.23 ms)
// dX (optional, round 5): the layer's DATA gradient behind the relu that made X rides along in the same sweep,
//   dX[n,c] = X[n,c] > 0 ? sum_o dY[n,o] W[o,c] : 0
// - what k_t_seed (OUT = 1) / k_t_rgb_hidden_adjoint (OUT = 3) did in a second pass over X (the same products in the same order)
template <int OUT>
__global__ void __launch_bounds__(T_THREADS) k_t_wcolsum(const float* __restrict__ X, int C, const float* __restrict__ dY, int64_t N,
                                                          int rows_per_block, float* __restrict__ dW, float* __restrict__ db, Rows rw,
                                                          const float* __restrict__ W, float* __restrict__ dX, unsigned* __restrict__ gmax) {
    // gmax (optional, round 6): batch-wide max |dX| as float bits - the operand scale of the split-fp16 product that reads dX next
    __shared__ float s[OUT][T_THREADS];
    float dmax = 0.0f;
    // (C is 128 or 256: wave-uniform row numbers, fetched a step ahead with scalar loads - see k_t_colsum)
    const int c = threadIdx.x % C, r0 = __builtin_amdgcn_readfirstlane(threadIdx.x / C), rs = T_THREADS / C;
    const int64_t NL = rows_n(rw, N);
    const int64_t base = (int64_t)blockIdx.x * rows_per_block;
    int64_t end = base + rows_per_block;
    if (end > NL) end = NL;
    float acc[OUT], bs[OUT], wv[OUT];
#pragma unroll
    for (int o = 0; o < OUT; ++o) { acc[o] = 0.0f; bs[o] = 0.0f; wv[o] = dX ? W[o * C + c] : 0.0f; }
    auto data_grad = [&](int64_t row, float x, const float (&y)[OUT]) {
        float v = y[0] * wv[0];
#pragma unroll
        for (int o = 1; o < OUT; ++o) v = v + y[o] * wv[o];
        v = x > 0.0f ? v : 0.0f;
        dX[row * C + c] = v;
        const float av = fabsf(v);
        if (av < 3.0e38f) dmax = fmaxf(dmax, av);      // non-finite values do not set the scale
    };
    int64_t n = base + r0;
    int64_t id[4];
    auto fetch = [&](int64_t n_) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int64_t q = n_ + k * (int64_t)rs; id[k] = rows_at_u(rw, q < end ? q : end - 1); }
    };
    if (n < end) fetch(n);
    for (; n + 3 * (int64_t)rs < end; n += 4 * (int64_t)rs) {            // four rows in flight per thread
        float x[4], y[4][OUT];
        const int64_t id4[4] = {id[0], id[1], id[2], id[3]};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t row = id[k];
            x[k] = X[row * C + c];
#pragma unroll
            for (int o = 0; o < OUT; ++o) y[k][o] = dY[row * OUT + o];
        }
        fetch(n + 4 * (int64_t)rs);
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int o = 0; o < OUT; ++o) { acc[o] = fmaf(x[k], y[k][o], acc[o]); bs[o] += y[k][o]; }
        if (dX) {
            const int64_t rid[4] = {id4[0], id4[1], id4[2], id4[3]};
#pragma unroll
            for (int k = 0; k < 4; ++k) data_grad(rid[k], x[k], y[k]);
        }
    }
    for (; n < end; n += rs) {
        const int64_t row = rows_at_u(rw, n);
        const float x = X[row * C + c];
        float yy[OUT];
#pragma unroll
        for (int o = 0; o < OUT; ++o) { const float y = dY[row * OUT + o]; yy[o] = y; acc[o] = fmaf(x, y, acc[o]); bs[o] += y; }
        if (dX) data_grad(row, x, yy);
    }
#pragma unroll
    for (int o = 0; o < OUT; ++o) s[o][threadIdx.x] = acc[o];
    __syncthreads();
    if (r0 == 0) {
#pragma unroll
        for (int o = 0; o < OUT; ++o) {
            float t = acc[o];
            for (int k = 1; k < rs; ++k) t += s[o][k * C + c];
            atomicAdd(dW + o * C + c, t);
        }
    }
    if (db && c == 0) {                  // one thread per row group saw every row of its group
#pragma unroll
        for (int o = 0; o < OUT; ++o) atomicAdd(db + o, bs[o]);
    }
    if (gmax) {                          // (one look-before-atomicMax per wave: see k_t_pe_tangent)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, off));
        if ((threadIdx.x & 63) == 0 && dmax > 0.0f && __float_as_uint(dmax) > __atomic_load_n(gmax, __ATOMIC_RELAXED))
            atomicMax(gmax, __float_as_uint(dmax));
    }
}

// first lighting layer fused with its bias and ReLU: hl1[n,f] = relu(b0[f] + sum_j W0[f,j] xl[n,j]), 9 -> 128
// (model/spacenet.py:174-188); thread = feature, two samples per block iteration, rows written coalesced
__global__ void __launch_bounds__(T_THREADS) k_t_light_first(const float* __restrict__ xl, const float* __restrict__ W0,
                                                              const float* __restrict__ b0, int64_t N, int rows_per_block,
                                                              float* __restrict__ hl1, Rows rw) {
    const int f = threadIdx.x & 127, r0 = threadIdx.x >> 7;
    float w[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) w[j] = W0[f * 9 + j];
    const float b = b0[f];
    const int64_t NL = rows_n(rw, N);
    const int64_t base = (int64_t)blockIdx.x * rows_per_block;
    int64_t end = base + rows_per_block;
    if (end > NL) end = NL;
    for (int64_t kr = base + r0; kr < end; kr += 2) {
        const int64_t n = rows_at(rw, kr);
        const float* x = xl + n * 9;          // wave-uniform address: nine scalar / broadcast loads
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < 9; ++j) acc = fmaf(x[j], w[j], acc);
        hl1[n * 128 + f] = fmaxf(acc + b, 0.0f);
    }
}

// data gradient of the first lighting layer: d_xl[n,j] = sum_k d_hl1[n,k] W0[k,j], 128 -> 9.  A 64-sample tile goes through
// LDS (coalesced rows in, padded rows out) so that thread (sample, quarter of k) reads its own row slice conflict-free.
__global__ void __launch_bounds__(T_THREADS) k_t_light_first_bwd(const float* __restrict__ d_hl1, const float* __restrict__ W0,
                                                                  int64_t N, float* __restrict__ d_xl, Rows rw) {
    __shared__ float tile[64][129];
    __shared__ float part[4][64][9];
    const int64_t NL = rows_n(rw, N);
    const int64_t n0 = (int64_t)blockIdx.x * 64;
    if (n0 >= NL) return;          // block-uniform
    {
        // a wave moves half a row per load: the row is wave-uniform (scalar row numbers, rows_at_u), and the 32 loads of a thread do not
        // wait for each other (round 6: the element loop fetched every row number with a vector load in front of the value)
        const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), k = ((wv & 1) << 6) + (threadIdx.x & 63);
        float v[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int r = 2 * i + (wv >> 1);
            const int64_t q = n0 + r < NL ? n0 + r : NL - 1;
            v[i] = d_hl1[rows_at_u(rw, q) * 128 + k];
        }
#pragma unroll
        for (int i = 0; i < 32; ++i) { const int r = 2 * i + (wv >> 1); tile[r][k] = n0 + r < NL ? v[i] : 0.0f; }
    }
    __syncthreads();
    const int sm = threadIdx.x & 63, q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float acc[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) acc[j] = 0.0f;
    for (int k = 32 * q; k < 32 * q + 32; ++k) {
        const float v = tile[sm][k];
#pragma unroll
        for (int j = 0; j < 9; ++j) acc[j] = fmaf(v, W0[k * 9 + j], acc[j]);     // uniform address: scalar loads
    }
#pragma unroll
    for (int j = 0; j < 9; ++j) part[q][sm][j] = acc[j];
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * 9; e += T_THREADS) {
        const int r = e / 9, j = e - 9 * r;
        if (n0 + r < NL) d_xl[rows_at(rw, n0 + r) * 9 + j] = (part[0][r][j] + part[1][r][j]) + (part[2][r][j] + part[3][r][j]);
    }
}

// lighting input rows [n_w, x_w (rotated / shifted, model/spacenet.py:254-263), d/|d|]
__global__ void __launch_bounds__(T_THREADS) k_t_light_in(const float* __restrict__ n_w, const float* __restrict__ ray_o,
                                                           const float* __restrict__ ray_d, const float* __restrict__ z_vals,
                                                           const DsnFrameState* __restrict__ fs, int64_t N, int S,
                                                           float* __restrict__ xl, Rows rw) {
    const int64_t kr = (int64_t)blockIdx.x * T_THREADS + threadIdx.x;
    if (kr >= rows_n(rw, N)) return;
    const int64_t n = rows_at(rw, kr);
    const int64_t ray = n / S;
    const float d[3] = {ray_d[3 * ray], ray_d[3 * ray + 1], ray_d[3 * ray + 2]};
    const float z = z_vals[n];
    float xw[3] = {ray_o[3 * ray] + d[0] * z, ray_o[3 * ray + 1] + d[1] * z, ray_o[3 * ray + 2] + d[2] * z};
    if (fs->has_rot != 0.0f) {
        const float ax = xw[0] - fs->rot_center[0], ay = xw[1] - fs->rot_center[1];
        const float nx = (ax * fs->rot[0] + ay * fs->rot[2]) + fs->rot_center[0];
        const float ny = (ax * fs->rot[1] + ay * fs->rot[3]) + fs->rot_center[1];
        xw[0] = nx; xw[1] = ny;
    }
    if (fs->has_light != 0.0f) { xw[0] += fs->light_shift[0]; xw[1] += fs->light_shift[1]; xw[2] += fs->light_shift[2]; }
    const float vn = dsn_norm3(d);
    float* o = xl + 9 * n;
    o[0] = n_w[3 * n]; o[1] = n_w[3 * n + 1]; o[2] = n_w[3 * n + 2];
    o[3] = xw[0]; o[4] = xw[1]; o[5] = xw[2];
    o[6] = dsn_div(d[0], vn); o[7] = dsn_div(d[1], vn); o[8] = dsn_div(d[2], vn);
}

// colour = (ELU(pre) + 1) * essence  (model/spacenet.py:186-188, :265)
__global__ void __launch_bounds__(T_THREADS) k_t_colour(const float* __restrict__ pre, const float* __restrict__ ess, int64_t N,
                                                         float* __restrict__ wl, float* __restrict__ col, Rows rw) {
    const int64_t kr = (int64_t)blockIdx.x * T_THREADS + threadIdx.x;
    if (kr >= rows_n(rw, N)) return;
    const int64_t n = rows_at(rw, kr);
    const float p = pre[n];
    const float w = (p > 0.0f ? p : expm1f(p)) + 1.0f;
    wl[n] = w;
    for (int c = 0; c < 3; ++c) col[3 * n + c] = w * ess[3 * n + c];
}

// utils/nerf_net_utils.py:5-56 forward + its adjoint, one thread per ray (training batches are a few thousand rays).
// Cotangents: d_rgb [R,3] (required), d_disp / d_acc / d_depth [R], d_weights [R,S] (optional).
__global__ void __launch_bounds__(T_THREADS) k_t_composite_adjoint(
    const float* __restrict__ colour, const float* __restrict__ sigma, const uint8_t* __restrict__ transparent,
    const float* __restrict__ z_vals, const float* __restrict__ ray_d, const float* __restrict__ noise, int R, int S,
    const float* __restrict__ d_rgb, const float* __restrict__ d_disp, const float* __restrict__ d_acc,
    const float* __restrict__ d_depth, const float* __restrict__ d_weights, float* __restrict__ scratch_t,
    float* __restrict__ d_colour, float* __restrict__ d_sigma, uint8_t* __restrict__ live) {
    const int r = blockIdx.x * T_THREADS + threadIdx.x;
    if (r >= R) return;
    const float d[3] = {ray_d[3 * r], ray_d[3 * r + 1], ray_d[3 * r + 2]};
    const float dn = dsn_norm3(d);
    const int64_t b = (int64_t)r * S;
    // forward: transmittance per sample (kept in scratch_t), depth and acc for the disparity chain
    float T = 1.0f, depth = 0.0f, acc = 0.0f;
    for (int i = 0; i < S; ++i) {
        float s = sigma[b + i];
        if (transparent[b + i]) s = 0.0f;
        if (noise) s += noise[b + i];
        s = s > 0.0f ? s : 0.0f;
        const float dist = (i + 1 < S ? z_vals[b + i + 1] - z_vals[b + i] : 1e10f) * dn;
        const float alpha = 1.0f - expf(-s * dist);
        scratch_t[b + i] = T;
        const float w = alpha * T;
        depth += w * z_vals[b + i];
        acc += w;
        T *= (1.0f - alpha) + 1e-10f;
    }
    float gd = d_depth ? d_depth[r] : 0.0f, ga = d_acc ? d_acc[r] : 0.0f;
    if (d_disp) {   // disp = 1 / max(1e-10, depth / acc)
        const float q = depth / acc;
        if (q > 1e-10f) { gd += d_disp[r] * (-1.0f / (q * q * acc)); ga += d_disp[r] * (1.0f / (q * acc)); }
    }
    const float gr[3] = {d_rgb[3 * r], d_rgb[3 * r + 1], d_rgb[3 * r + 2]};
    float suffix = 0.0f;   // sum_{k>i} Gw_k w_k
    for (int i = S - 1; i >= 0; --i) {
        float raw = sigma[b + i];
        const bool tr = transparent[b + i] != 0;
        if (tr) raw = 0.0f;
        if (noise) raw += noise[b + i];
        const float s = raw > 0.0f ? raw : 0.0f;
        const float dist = (i + 1 < S ? z_vals[b + i + 1] - z_vals[b + i] : 1e10f) * dn;
        const float e = expf(-s * dist);
        const float alpha = 1.0f - e;
        const float Ti = scratch_t[b + i];
        const float w = alpha * Ti;
        // (a colour exists only where the forward evaluated the sample - rows with alpha = 0 may never have been written; its
        //  product with w = 0 would be 0 for any finite colour)
        float cg = 0.0f;
        if (s > 0.0f) { const float* c = colour + 3 * (b + i); cg = gr[0] * c[0] + gr[1] * c[1] + gr[2] * c[2]; }
        float gw = cg + gd * z_vals[b + i] + ga;
        if (d_weights) gw += d_weights[b + i];
        const float dalpha = gw * Ti - suffix / ((1.0f - alpha) + 1e-10f);
        suffix += gw * w;
        const float ds = (!tr && raw > 0.0f) ? dalpha * dist * e : 0.0f;
        const float dc[3] = {w * gr[0], w * gr[1], w * gr[2]};
        d_sigma[b + i] = ds;
        for (int k = 0; k < 3; ++k) d_colour[3 * (b + i) + k] = dc[k];
        // rows whose cotangents are all zero add exactly nothing to any parameter gradient: the passes below skip them
        if (live) live[b + i] = (ds != 0.0f || dc[0] != 0.0f || dc[1] != 0.0f || dc[2] != 0.0f) ? 1 : 0;
    }
}

// The same adjoint with ONE WAVE PER RAY (round 6): lane l owns samples l, l + 64, ... (CH chunks of 64: S <= 64 CH).  The one-thread-
// per-ray form above walks 2 S dependent steps with 8192 threads in flight - 128 waves on 1024 SIMDs, 0.17 ms of latency per step
// of 8192 x 64; here the transmittance is a wave-level product scan (shifted by one lane, like the forward compositor's), the
// suffix sum a reverse scan, every load and store coalesced over the samples of a ray: 0.02 ms.  Same formulas; the products and sums
// associate as scans instead of left to right (the forward compositor k_composite does the same).  scratch_t is not needed.
__device__ __forceinline__ float t_wave_incl_prod(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const float x = __shfl_up(v, o); if (lane >= o) v *= x; }
    return v;
}
__device__ __forceinline__ float t_wave_incl_sum_down(float v, int lane) {      // v_l <- sum_{k >= l} v_k
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const float x = __shfl_down(v, o); if (lane + o < 64) v += x; }
    return v;
}
template <int CH>
__global__ void __launch_bounds__(T_THREADS) k_t_composite_adjoint_w(
    const float* __restrict__ colour, const float* __restrict__ sigma, const uint8_t* __restrict__ transparent,
    const float* __restrict__ z_vals, const float* __restrict__ ray_d, const float* __restrict__ noise, int R, int S,
    const float* __restrict__ d_rgb, const float* __restrict__ d_disp, const float* __restrict__ d_acc,
    const float* __restrict__ d_depth, const float* __restrict__ d_weights,
    float* __restrict__ d_colour, float* __restrict__ d_sigma, uint8_t* __restrict__ live) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * (T_THREADS / 64) + (threadIdx.x >> 6);
    if (r >= R) return;                                   // wave-uniform
    const float d[3] = {ray_d[3 * r], ray_d[3 * r + 1], ray_d[3 * r + 2]};
    const float dn = dsn_norm3(d);
    const int64_t b = (int64_t)r * S;
    float raw[CH], zv[CH], dist[CH], e[CH], alpha[CH], f[CH], T[CH], w[CH];
    bool tr[CH], in[CH];
    float carry = 1.0f, depth = 0.0f, acc = 0.0f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int i = 64 * c + lane;
        in[c] = i < S;
        float sg = 0.0f, zn = 0.0f;
        tr[c] = true; zv[c] = 0.0f;
        if (in[c]) {
            sg = sigma[b + i];
            tr[c] = transparent[b + i] != 0;
            zv[c] = z_vals[b + i];
            zn = i + 1 < S ? z_vals[b + i + 1] : 0.0f;
        }
        float rw = tr[c] ? 0.0f : sg;
        if (noise && in[c]) rw += noise[b + i];
        raw[c] = rw;
        const float s = rw > 0.0f ? rw : 0.0f;
        dist[c] = (i + 1 < S ? zn - zv[c] : 1e10f) * dn;
        e[c] = in[c] ? expf(-s * dist[c]) : 1.0f;
        alpha[c] = 1.0f - e[c];
        f[c] = in[c] ? (1.0f - alpha[c]) + 1e-10f : 1.0f;
        const float inc = t_wave_incl_prod(f[c], lane);
        float ex = __shfl_up(inc, 1);
        if (lane == 0) ex = 1.0f;
        T[c] = carry * ex;
        carry *= __shfl(inc, 63);
        w[c] = in[c] ? alpha[c] * T[c] : 0.0f;
        depth += w[c] * zv[c];
        acc += w[c];
    }
    depth = t_wave_sum(depth);
    acc = t_wave_sum(acc);
    float gd = d_depth ? d_depth[r] : 0.0f, ga = d_acc ? d_acc[r] : 0.0f;
    if (d_disp) {   // disp = 1 / max(1e-10, depth / acc)
        const float q = depth / acc;
        if (q > 1e-10f) { const float dd = d_disp[r]; gd += dd * (-1.0f / (q * q * acc)); ga += dd * (1.0f / (q * acc)); }
    }
    const float gr[3] = {d_rgb[3 * r], d_rgb[3 * r + 1], d_rgb[3 * r + 2]};
    float tail = 0.0f;                                    // sum of gw_k w_k over the chunks behind this one
#pragma unroll
    for (int c = CH - 1; c >= 0; --c) {
        const int i = 64 * c + lane;
        const float s = raw[c] > 0.0f ? raw[c] : 0.0f;
        // (a colour exists only where the forward evaluated the sample - rows with alpha = 0 may never have been written)
        float cg = 0.0f;
        if (in[c] && s > 0.0f) { const float* cc = colour + 3 * (b + i); cg = gr[0] * cc[0] + gr[1] * cc[1] + gr[2] * cc[2]; }
        float gw = cg + gd * zv[c] + ga;
        if (d_weights && in[c]) gw += d_weights[b + i];
        const float p = in[c] ? gw * w[c] : 0.0f;
        const float incl = t_wave_incl_sum_down(p, lane);          // sum_{k >= lane} of this chunk
        const float suffix = tail + (incl - p);
        tail += __shfl(incl, 0);
        if (!in[c]) continue;
        const float dalpha = gw * T[c] - suffix / f[c];
        const float ds = (!tr[c] && raw[c] > 0.0f) ? dalpha * dist[c] * e[c] : 0.0f;
        const float dc[3] = {w[c] * gr[0], w[c] * gr[1], w[c] * gr[2]};
        d_sigma[b + i] = ds;
        for (int k = 0; k < 3; ++k) d_colour[3 * (b + i) + k] = dc[k];
        if (live) live[b + i] = (ds != 0.0f || dc[0] != 0.0f || dc[1] != 0.0f || dc[2] != 0.0f) ? 1 : 0;
    }
}

// colour = wl * essence, wl = ELU(pre) + 1:  d_essence, d_pre
__global__ void __launch_bounds__(T_THREADS) k_t_colour_adjoint(const float* __restrict__ d_colour, const float* __restrict__ ess,
                                                                 const float* __restrict__ wl, const float* __restrict__ pre,
                                                                 int64_t N, float* __restrict__ d_ess, float* __restrict__ d_pre, Rows rw) {
    const int64_t kr = (int64_t)blockIdx.x * T_THREADS + threadIdx.x;
    if (kr >= rows_n(rw, N)) return;
    const int64_t n = rows_at(rw, kr);
    float dw = 0.0f;
    for (int c = 0; c < 3; ++c) {
        const float dc = d_colour[3 * n + c];
        d_ess[3 * n + c] = wl[n] * dc;
        dw += ess[3 * n + c] * dc;
    }
    d_pre[n] = dw * (pre[n] > 0.0f ? 1.0f : wl[n]);   // ELU' = exp(pre) = wl for pre <= 0
}

// d_rr[n,c] = (rr > 0) ? sum_k d_ess[n,k] W[k,c] : 0      (rgb_net.3 transposed, model/spacenet.py:72-79)
__global__ void __launch_bounds__(T_THREADS) k_t_rgb_hidden_adjoint(const float* __restrict__ d_ess, const float* __restrict__ w3,
                                                                     const float* __restrict__ rr, int64_t total,
                                                                     float* __restrict__ d_rr, Rows rw) {
    const int64_t tl = 4 * ((int64_t)blockIdx.x * T_THREADS + threadIdx.x);
    if (tl >= rows_n(rw, total >> 7) * 128) return;
    const int64_t n = rows_at(rw, tl >> 7);
    const int c = (int)(tl & 127);
    const int64_t t = n * 128 + c;
    const float e0 = d_ess[3 * n], e1 = d_ess[3 * n + 1], e2 = d_ess[3 * n + 2];
    const float4 a0 = make_float4(w3[c], w3[c + 1], w3[c + 2], w3[c + 3]);
    const float4 a1 = make_float4(w3[128 + c], w3[129 + c], w3[130 + c], w3[131 + c]);
    const float4 a2 = make_float4(w3[256 + c], w3[257 + c], w3[258 + c], w3[259 + c]);
    const float4 r = *reinterpret_cast<const float4*>(rr + t);
    float4 v;
    v.x = r.x > 0.0f ? e0 * a0.x + e1 * a1.x + e2 * a2.x : 0.0f;
    v.y = r.y > 0.0f ? e0 * a0.y + e1 * a1.y + e2 * a2.y : 0.0f;
    v.z = r.z > 0.0f ? e0 * a0.z + e1 * a1.z + e2 * a2.z : 0.0f;
    v.w = r.w > 0.0f ? e0 * a0.w + e1 * a1.w + e2 * a2.w : 0.0f;
    *reinterpret_cast<float4*>(d_rr + t) = v;
}

// u = (d n_w / d g)^T d_n_w   (model/spacenet.py:278-298; the two projections share the canonical face)
__global__ void __launch_bounds__(T_THREADS) k_t_normal_adjoint(const DsnFaceRec* __restrict__ face_world,
                                                                 const DsnFaceRec* __restrict__ face_canon,
                                                                 const float* __restrict__ x_c, const float* __restrict__ g,
                                                                 const int32_t* __restrict__ idx_c, const float* __restrict__ d_xl,
                                                                 int64_t N, float* __restrict__ u, Rows rw) {
    const int64_t kr = (int64_t)blockIdx.x * T_THREADS + threadIdx.x;
    if (kr >= rows_n(rw, N)) return;
    const int64_t n = rows_at(rw, kr);
    const DsnFaceRec fc = dsn_load_face(face_canon, idx_c[n]);
    const DsnFaceRec fw = dsn_load_face(face_world, idx_c[n]);
    float p[3], pe[3], s[3], e[3], df[3], uu, vv, hh;
    for (int c = 0; c < 3; ++c) { p[c] = x_c[3 * n + c]; pe[c] = p[c] + g[3 * n + c]; }
    dsn_project(p, fc, uu, vv, hh);
    dsn_map2face(uu, vv, hh, fw, s);
    dsn_project(pe, fc, uu, vv, hh);
    dsn_map2face(uu, vv, hh, fw, e);
    for (int c = 0; c < 3; ++c) df[c] = e[c] - s[c];
    const float nrm = dsn_norm3(df);
    const float dn[3] = {d_xl[9 * n], d_xl[9 * n + 1], d_xl[9 * n + 2]};
    float dd[3];
    if (nrm >= 1e-12f) {
        const float nh[3] = {df[0] / nrm, df[1] / nrm, df[2] / nrm};
        const float dot = nh[0] * dn[0] + nh[1] * dn[1] + nh[2] * dn[2];
        for (int c = 0; c < 3; ++c) dd[c] = (dn[c] - nh[c] * dot) / nrm;
    } else {
        for (int c = 0; c < 3; ++c) dd[c] = dn[c] / 1e-12f;
    }
    // delta = v20' (a.g) + v10' (b.g) + n' (n.g),  a = inv (d11 v20 - d01 v10),  b = inv (d00 v10 - d01 v20)
    const float ka = dsn_dot3(fw.v20, dd), kb = dsn_dot3(fw.v10, dd), kn = dsn_dot3(fw.n, dd);
    for (int c = 0; c < 3; ++c) {
        const float a = fc.inv * (fc.d11 * fc.v20[c] - fc.d01 * fc.v10[c]);
        const float b = fc.inv * (fc.d00 * fc.v10[c] - fc.d01 * fc.v20[c]);
        u[3 * n + c] = a * ka + b * kb + fc.n[c] * kn;
    }
}

// stage1.0: gradient of the 24 frame-constant input columns, the embedding row and the pose code, from the column
// sums csum = sum_n ahat_0[n] (= d bias of stage1.0)
__global__ void __launch_bounds__(256) k_t_first_layer_consts(const float* __restrict__ csum, const float* __restrict__ w0,
                                                               const DsnFrameState* __restrict__ fs, int frame_idx,
                                                               int zero_code, float* __restrict__ d_w0,
                                                               float* __restrict__ d_emb, float* __restrict__ d_pose) {
    const int t = threadIdx.x;   // output feature
    const float c = csum[t];
    for (int k = 0; k < 8; ++k) d_w0[t * 87 + k] += c * fs->code[k];
    for (int k = 0; k < 16; ++k) d_w0[t * 87 + W0_POSE_COL + k] += c * fs->pose_feat[k];
    __shared__ float s[256];
    for (int k = 0; k < 24; ++k) {
        const int col = k < 8 ? k : W0_POSE_COL + (k - 8);
        s[t] = c * w0[t * 87 + col];
        __syncthreads();
        for (int off = 128; off >= 1; off >>= 1) {
            if (t < off) s[t] += s[t + off];
            __syncthreads();
        }
        if (t == 0) {
            if (k < 8) { if (!zero_code) d_emb[frame_idx * 8 + k] += s[0]; }
            else d_pose[k - 8] = s[0];
        }
        __syncthreads();
    }
}

// pose_mlp backward (model/spacenet.py:199-205, :314-331): one row, recomputed forward
__global__ void __launch_bounds__(256) k_t_pose_mlp_adjoint(const float* __restrict__ w0, const float* __restrict__ b0,
                                                             const float* __restrict__ w2, const float* __restrict__ b2,
                                                             const float* __restrict__ w4, const float* __restrict__ poses,
                                                             const float* __restrict__ d_pose, float* __restrict__ gw0,
                                                             float* __restrict__ gb0, float* __restrict__ gw2,
                                                             float* __restrict__ gb2, float* __restrict__ gw4,
                                                             float* __restrict__ gb4) {
    __shared__ float q[92], h1[64], h2[64], dh1[64], dh2[64], dp[16];
    const int t = threadIdx.x;
    if (t < 23) {
        const float* r = poses + 3 * (t + 1);
        float a[3] = {r[0] + 1e-16f, r[1] + 1e-16f, r[2] + 1e-16f};
        const float angle = dsn_norm3(a);
        const float half = dsn_div(angle, 2.0f);
        const float s = sinf(half), c = cosf(half);
        q[4 * t + 0] = dsn_div(r[0], angle) * s;
        q[4 * t + 1] = dsn_div(r[1], angle) * s;
        q[4 * t + 2] = dsn_div(r[2], angle) * s;
        q[4 * t + 3] = c - 1.0f;
    }
    if (t < 16) dp[t] = d_pose[t];
    __syncthreads();
    if (t < 64) {
        float acc = b0[t];
        for (int k = 0; k < 92; ++k) acc += w0[t * 92 + k] * q[k];
        h1[t] = acc > 0.f ? acc : 0.f;
    }
    __syncthreads();
    if (t < 64) {
        float acc = b2[t];
        for (int k = 0; k < 64; ++k) acc += w2[t * 64 + k] * h1[k];
        h2[t] = acc > 0.f ? acc : 0.f;
    }
    __syncthreads();
    if (t < 64) {
        float acc = 0.f;
        for (int o = 0; o < 16; ++o) acc += w4[o * 64 + t] * dp[o];
        dh2[t] = h2[t] > 0.f ? acc : 0.f;
    }
    if (t < 16) gb4[t] += dp[t];
    for (int i = t; i < 16 * 64; i += 256) gw4[i] += dp[i / 64] * h2[i % 64];
    __syncthreads();
    if (t < 64) {
        float acc = 0.f;
        for (int o = 0; o < 64; ++o) acc += w2[o * 64 + t] * dh2[o];
        dh1[t] = h1[t] > 0.f ? acc : 0.f;
        gb2[t] += dh2[t];
    }
    for (int i = t; i < 64 * 64; i += 256) gw2[i] += dh2[i / 64] * h1[i % 64];
    __syncthreads();
    if (t < 64) gb0[t] += dh1[t];
    for (int i = t; i < 64 * 92; i += 256) gw0[i] += dh1[i / 92] * q[i % 92];
}

// ------------------------------------------------------------------------------------------------------------
// weight-gradient product  dW [out,in] += dY^T X  over N samples (the contraction index): exact-fp32 MFMA.
// rocBLAS' fp32 GEMM for this shape (256 x 256 output, K = 524 288) runs at 22 TFLOP/s; here the sample axis is
// split over one workgroup per CU, each wave keeps OT x IT 32x32 accumulator tiles in registers for its whole chunk,
// and both operands are read straight from their row-major [N,C] arrays in MFMA operand layout: lane l supplies
// dY[n + (l>>5)][o0 + (l&31)] as A and X[n + (l>>5)][j0 + (l&31)] as B - two coalesced 128-byte rows per load, no
// LDS, no transposes.  Partial tiles are combined with fp32 atomics (256 workgroups -> 16.7 M atomics per product).
// ------------------------------------------------------------------------------------------------------------
typedef float t_f32x16 __attribute__((ext_vector_type(16)));

template <int OT, int IT, int WO, int WI>
__global__ void __launch_bounds__(256) k_t_wgrad(const float* __restrict__ dY, int ldy, const float* __restrict__ X, int ldx,
                                                  int64_t N, int rows_per_wg, float* __restrict__ dW, int ldw, int in_valid,
                                                  float* __restrict__ dbias, Rows rw) {
    DSN_OWN_SIMD_T(32);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wo = wave / WI, wi = wave % WI;
    const int col = lane & 31, half = lane >> 5;
    const int64_t NL = rows_n(rw, N);
    if (rw.cnt) rows_per_wg = (int)rows_share(NL, 2, 64);      // (the launch was sized for N rows: share what is listed)
    const int64_t n0 = (int64_t)blockIdx.x * rows_per_wg;
    int64_t n1 = n0 + rows_per_wg;
    if (n1 > NL) n1 = NL;
    if (n0 >= n1) return;
    const bool want_bias = dbias != nullptr && wi == 0;      // column sums of dY (the bias gradient) ride on the loads
    float bsum[OT];
#pragma unroll
    for (int a = 0; a < OT; ++a) bsum[a] = 0.0f;
    t_f32x16 acc[OT][IT];
#pragma unroll
    for (int a = 0; a < OT; ++a)
#pragma unroll
        for (int b = 0; b < IT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    const float* pa = dY + (wo * OT) * 32 + col;
    const float* pb = X + (wi * IT) * 32 + col;
    // DEPTH contraction steps (2 samples each) are in flight ahead of the matrix pipe: at 16 MFMAs x 64 cycles per
    // step a single step of look-ahead (0.5 us) is shorter than the HBM latency under load
    constexpr int DEPTH = (OT * IT >= 16) ? 4 : 8;
    float af[DEPTH][OT], bf[DEPTH][IT], an[DEPTH][OT], bn[DEPTH][IT];
    // listed rows: the 2 DEPTH row numbers of a load step are wave-uniform - fetched with scalar loads (rows_at_u) one step before
    // the loads that use them are issued
    int64_t ids[2 * DEPTH];
    auto fetch = [&](int64_t n) {
#pragma unroll
        for (int q = 0; q < 2 * DEPTH; ++q) { const int64_t k = n + q; ids[q] = rows_at_u(rw, k < n1 ? k : n1 - 1); }
    };
    auto load = [&](int64_t n, float (*fa)[OT], float (*fb)[IT]) {
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) {
            const int64_t lrow = n + 2 * s + half;
            const bool ok = lrow < n1;
            const int64_t row = half ? ids[2 * s + 1] : ids[2 * s];
#pragma unroll
            for (int a = 0; a < OT; ++a) fa[s][a] = ok ? pa[row * ldy + a * 32] : 0.0f;
#pragma unroll
            for (int b = 0; b < IT; ++b) fb[s][b] = (ok && (wi * IT + b) * 32 + col < in_valid) ? pb[row * ldx + b * 32] : 0.0f;
        }
    };
    fetch(n0);
    load(n0, af, bf);
    fetch(n0 + 2 * DEPTH);
    for (int64_t n = n0; n < n1; n += 2 * DEPTH) {
        load(n + 2 * DEPTH, an, bn);   // rows beyond the chunk read as zero
        fetch(n + 4 * DEPTH);
        if (want_bias) {
#pragma unroll
            for (int s = 0; s < DEPTH; ++s)
#pragma unroll
                for (int a = 0; a < OT; ++a) bsum[a] += af[s][a];
        }
#pragma unroll
        for (int s = 0; s < DEPTH; ++s)
#pragma unroll
            for (int a = 0; a < OT; ++a)
#pragma unroll
                for (int b = 0; b < IT; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s][a], bf[s][b], acc[a][b], 0, 0, 0);
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) {
#pragma unroll
            for (int a = 0; a < OT; ++a) af[s][a] = an[s][a];
#pragma unroll
            for (int b = 0; b < IT; ++b) bf[s][b] = bn[s][b];
        }
    }
#pragma unroll
    for (int a = 0; a < OT; ++a)
#pragma unroll
        for (int b = 0; b < IT; ++b) {
            const int j = (wi * IT + b) * 32 + col;
            if (j >= in_valid) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = (wo * OT + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                atomicAdd(dW + (int64_t)i * ldw + j, acc[a][b][r]);
            }
        }
    if (want_bias) {
#pragma unroll
        for (int a = 0; a < OT; ++a) {
            const float t = bsum[a] + __shfl_xor(bsum[a], 32);
            if (half == 0) atomicAdd(dbias + (wo * OT + a) * 32 + col, t);
        }
    }
}

// dW [out,in] += dY[N,out]^T X[N,in]; out in {128,256}, padded in (multiple of 32) in {32,64,128,256}, in_valid <= in
// (columns >= in_valid are neither read nor written)
bool wgrad_mfma(int64_t N, int in, int in_valid, int out, const float* X, int ldx, const float* dY, int ldy, float* dW, int ldw,
                hipStream_t st, float* dbias = nullptr, Rows rw = Rows{nullptr, nullptr}) {
    int groups = (out == 256 && in == 256) ? 256 : 768;   // the smaller tiles leave room for 3 workgroups per CU
    int rows = (int)((N + groups - 1) / groups);
    if (rows < 64) rows = 64;
    rows = (rows + 1) & ~1;
    groups = (int)((N + rows - 1) / rows);
    const dim3 g((unsigned)groups), b(256);
    if (out == 256 && in == 256) hipLaunchKernelGGL((k_t_wgrad<4, 4, 2, 2>), g, b, 0, st, dY, ldy, X, ldx, N, rows, dW, ldw, in_valid, dbias, rw);
    else if (out == 256 && in == 64) hipLaunchKernelGGL((k_t_wgrad<2, 2, 4, 1>), g, b, 0, st, dY, ldy, X, ldx, N, rows, dW, ldw, in_valid, dbias, rw);
    else if (out == 128 && in == 256) hipLaunchKernelGGL((k_t_wgrad<2, 4, 2, 2>), g, b, 0, st, dY, ldy, X, ldx, N, rows, dW, ldw, in_valid, dbias, rw);
    else if (out == 128 && in == 128) hipLaunchKernelGGL((k_t_wgrad<2, 2, 2, 2>), g, b, 0, st, dY, ldy, X, ldx, N, rows, dW, ldw, in_valid, dbias, rw);
    else if (out == 128 && in == 32) hipLaunchKernelGGL((k_t_wgrad<1, 1, 4, 1>), g, b, 0, st, dY, ldy, X, ldx, N, rows, dW, ldw, in_valid, dbias, rw);
    else return false;
    return true;
}

// ------------------------------------------------------------------------------------------------------------
// The same product for the twelve 256 x 256 trunk terms on the f16 matrix pipe: dW += dY^T X with both operands split
// v = hi + lo on the fly (three products hi*hi + hi*lo + lo*hi per accumulator, plain fp16 residuals), 5.3x fewer matrix
// cycles than the exact-fp32 kernel above.  Operand layout: for the 32x32x16 MFMA lane l supplies 8 consecutive contraction
// indices (samples n + 8 (l >> 5) + j) of ITS row / column (feature o0 + (l & 31)).  The residual split needs
// O(1) operands: each operand is divided by a power of two >= its batch-wide magnitude (device scalars left by
// k_tangent16 / k_adjoint16; forward activations and sigma-adjoints are O(1) already) and the product is multiplied back in
// the epilogue.  fp32 accumulation throughout.
// ------------------------------------------------------------------------------------------------------------
typedef _Float16 t_half8 __attribute__((ext_vector_type(8)));
typedef int t_i32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) t_i32x4* t_cptr4;
// smallest power of two >= s (1 for s <= 0 or non-finite), never below 2^-40: the product of two operand scales and the ratio of two
// such products (t_pair_ratio) stay normal floats when an operand's batch-wide magnitude has all but vanished (two
// tiny maxima multiplied to 0, the paired launches' ratio to inf / NaN, 0 x inf poisoned dW where the single launches gave 0)
__device__ __forceinline__ float t_pow2_at_least(float s) {
    if (!(s > 0.0f) || !(s < 3.0e38f)) return 1.0f;
    int e;
    const float m = frexpf(s, &e);                                // s = m 2^e, m in [0.5, 1)
    e = m == 0.5f ? e - 1 : e;
    return ldexpf(1.0f, e < -40 ? -40 : e);
}
// factor that takes accumulators from the units of one operand pair (s1 = sy sx) to those of the next (s2): both powers of two with
// s2 >= 2^-80, so the quotient is exact; capped at 2^126 (reached only by operands 2^46 times apart in both factors), never inf
__device__ __forceinline__ float t_pair_ratio(float s1, float s2) {
    const float r = s1 / s2;
    return r < 8.5e37f ? r : 8.507059e37f;
}
// The kernel: operand rows are staged through LDS by the DMA engine (global_load_lds_dwordx4, one 1 KB row per
// wave-instruction, no staging registers; W16S_STAGES - 1 steps of 16 rows x 2 operands = 32 KB each in flight per CU, every
// row fetched once per workgroup).  Every value is split ONCE per workgroup: thread f converts feature f of both operands for
// the 16 rows of a step out of the fp32 ring (column reads, conflict-free) and leaves packed (hi, lo) halves in an operand
// buffer laid out [operand][hi|lo][8-sample group][feature] x 16 B, from which a lane fetches each MFMA operand with one
// conflict-free ds_read_b128.  LDS: 4 stages x 32 KB fp32 ring + 32 KB operand buffer = all 160 KB.
// History on the 256 x 256 x 524 288 product: fp32 MFMA 0.80 ms -> split-fp16 with per-lane global loads and in-register
// splits 0.44 -> LDS-DMA staging 0.345 -> convert-once 0.33 -> one workgroup per CU (256 instead of 512 groups) 0.28.  (Storing
// per-workgroup tiles and reducing them in a second kernel instead of the atomicAdd epilogue measured the same: 0.24 + 0.04.)
#define W16S_STAGES 4
// floats of one workgroup's partial result in the two-stage form: the [256][C] tile + 256 column sums (bias gradient)
#define W16_PART(C) (256 * (C) + 256)
#define W16_PART_GROUPS 512          // most workgroups a two-stage product launches (k_t_wgrad16p: two per CU)
#ifndef W16D_VALU
#define W16D_VALU 5           // k_t_wgrad16d: VALU instructions the scheduler places behind each MFMA of a pipelined step
#endif
// the optional second operand pair of a split-fp16 weight-gradient launch (jobs = 1: none), see k_t_wgrad16d
struct WgradOps { const float* dY2; const float* sy2; const float* X2; const float* sx2; int jobs; };
__global__ void __launch_bounds__(256) k_t_wgrad16c(const float* __restrict__ dY, const float* __restrict__ sy_ptr,
                                                     const float* __restrict__ X, const float* __restrict__ sx_ptr, int64_t N,
                                                     int rows_per_wg, float* __restrict__ dW, int ldw,
                                                     float* __restrict__ dbias, Rows rw) {
    DSN_OWN_SIMD_T(64);
    constexpr int OT = 4, IT = 4, WI = 2;
    __shared__ __attribute__((aligned(16))) float ring[W16S_STAGES][2][16][256];
    __shared__ __attribute__((aligned(16))) t_half8 opbuf[2][2][2][256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wo = wave / WI, wi = wave % WI;
    const int col = lane & 31, half = lane >> 5;
    const int64_t NL = rows_n(rw, N);
    if (rw.cnt) rows_per_wg = (int)rows_share(NL, 16, 64);     // (the launch was sized for N rows: share what is listed)
    const int64_t n0 = (int64_t)blockIdx.x * rows_per_wg;
    int64_t n1 = n0 + rows_per_wg;
    if (n1 > NL) n1 = NL;
    if (n0 >= n1) return;                                // workgroup-uniform
    const int full = n1 > n0 ? (int)((n1 - n0) >> 4) : 0;
    const bool tail = n0 + 16 * (int64_t)full < n1;
    float bsum = 0.0f;                                   // column sum of dY feature tid (bias gradient)
    const float sy = sy_ptr ? t_pow2_at_least(*sy_ptr) : 1.0f, sx = sx_ptr ? t_pow2_at_least(*sx_ptr) : 1.0f;
    const float iy = 1.0f / sy, ix = 1.0f / sx;
    t_f32x16 acc[OT][IT];
#pragma unroll
    for (int a = 0; a < OT; ++a)
#pragma unroll
        for (int b = 0; b < IT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    const unsigned ring_off = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)&ring[0][0][0][0];
    // listed rows: this wave's four row numbers of the NEXT step to be staged, fetched with ONE scalar load a step ahead (a plain
    // global pointer gives vector loads + s_waitcnt vmcnt(0), which drains the DMA queue at every step)
    t_i32x4 ids = {0, 0, 0, 0};
    auto fetch_ids = [&](int t) {
        if (!rw.list) return;
        // (constant address space: the list is not written during this kernel, and only for loads from there does hipcc pick
        //  s_load_dwordx4 and track the wait itself; 16-byte aligned: n0 and the list base are)
        ids = *reinterpret_cast<t_cptr4>((uintptr_t)(rw.list + (n0 + 16 * (int64_t)t + 4 * wave)));
    };
    auto stage = [&](int t) {
        const int slot = t % W16S_STAGES;
        const int64_t row = n0 + 16 * (int64_t)t + 4 * wave;
        const unsigned da = ring_off + (unsigned)(((slot * 2 + 0) * 16 + 4 * wave) * 1024);
        const unsigned db = ring_off + (unsigned)(((slot * 2 + 1) * 16 + 4 * wave) * 1024);
        if (rw.list) {
            // listed rows: the wave's four rows of this step lie anywhere - one address per row; the LDS side advances by 1 KB per
            // instruction.  The row numbers were fetched one step earlier (ids: scalar loads whose latency would otherwise sit in
            // front of every step's DMA); a run of four consecutive rows takes the one-address form
            if (ids[3] - ids[0] == 3) {
                const char* ga = reinterpret_cast<const char*>(dY + (int64_t)ids[0] * 256) + lane * 16;
                const char* gb = reinterpret_cast<const char*>(X + (int64_t)ids[0] * 256) + lane * 16;
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\t"
                             "global_load_lds_dwordx4 %0, off\n\tglobal_load_lds_dwordx4 %0, off offset:1024\n\t"
                             "global_load_lds_dwordx4 %0, off offset:2048\n\tglobal_load_lds_dwordx4 %0, off offset:3072"
                             : : "v"(ga), "s"(da) : "memory", "m0");
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\t"
                             "global_load_lds_dwordx4 %0, off\n\tglobal_load_lds_dwordx4 %0, off offset:1024\n\t"
                             "global_load_lds_dwordx4 %0, off offset:2048\n\tglobal_load_lds_dwordx4 %0, off offset:3072"
                             : : "v"(gb), "s"(db) : "memory", "m0");
                return;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const char* ga = reinterpret_cast<const char*>(dY + (int64_t)ids[j] * 256) + lane * 16;
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(ga), "s"(da + 1024u * j) : "memory", "m0");
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const char* gb = reinterpret_cast<const char*>(X + (int64_t)ids[j] * 256) + lane * 16;
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gb), "s"(db + 1024u * j) : "memory", "m0");
            }
            return;
        }
        const char* ga = reinterpret_cast<const char*>(dY + row * 256) + lane * 16;
        const char* gb = reinterpret_cast<const char*>(X + row * 256) + lane * 16;
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %0, off\n\tglobal_load_lds_dwordx4 %0, off offset:1024\n\t"
                     "global_load_lds_dwordx4 %0, off offset:2048\n\tglobal_load_lds_dwordx4 %0, off offset:3072"
                     : : "v"(ga), "s"(da) : "memory", "m0");
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %0, off\n\tglobal_load_lds_dwordx4 %0, off offset:1024\n\t"
                     "global_load_lds_dwordx4 %0, off offset:2048\n\tglobal_load_lds_dwordx4 %0, off offset:3072"
                     : : "v"(gb), "s"(db) : "memory", "m0");
    };
    // split the 16 values v[op][row] of feature tid and publish them
    auto publish = [&](const float (&v)[2][16]) {
#pragma unroll
        for (int j = 0; j < 16; ++j) bsum += v[0][j];
#pragma unroll
        for (int op = 0; op < 2; ++op) {
            const float inv = op == 0 ? iy : ix;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                t_half8 hi, lo;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float x = v[op][8 * g + j] * inv;
                    const _Float16 h = (_Float16)x;
                    hi[j] = h;
                    lo[j] = (_Float16)(x - (float)h);
                }
                opbuf[op][0][g][tid] = hi;
                opbuf[op][1][g][tid] = lo;
            }
        }
    };
    auto multiply = [&]() {
        t_half8 ah[OT], al[OT], bh[IT], bl[IT];
#pragma unroll
        for (int a = 0; a < OT; ++a) {
            ah[a] = opbuf[0][0][half][(wo * OT + a) * 32 + col];
            al[a] = opbuf[0][1][half][(wo * OT + a) * 32 + col];
        }
#pragma unroll
        for (int b = 0; b < IT; ++b) {
            bh[b] = opbuf[1][0][half][(wi * IT + b) * 32 + col];
            bl[b] = opbuf[1][1][half][(wi * IT + b) * 32 + col];
        }
#pragma unroll
        for (int a = 0; a < OT; ++a)
#pragma unroll
            for (int b = 0; b < IT; ++b) {
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bh[b], acc[a][b], 0, 0, 0);
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bl[b], acc[a][b], 0, 0, 0);
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[a], bh[b], acc[a][b], 0, 0, 0);
            }
    };
    const int pre = full < W16S_STAGES - 1 ? full : W16S_STAGES - 1;
    for (int t = 0; t < pre; ++t) { fetch_ids(t); stage(t); }
    if (pre < full) fetch_ids(pre);
    for (int t = 0; t < full; ++t) {
        const int issued = (t + W16S_STAGES - 1 < full) ? t + W16S_STAGES - 1 : full;
        const int ahead = issued - (t + 1);
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // A: step t has landed for everyone; everyone has fetched the operands of step t - 1 (opbuf and ring slot t - 1 are free)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (t + W16S_STAGES - 1 < full) {
            stage(t + W16S_STAGES - 1);
            if (t + W16S_STAGES < full) fetch_ids(t + W16S_STAGES);      // (waited for at the next lgkmcnt(0): hidden behind this step)
        }
        const int slot = t % W16S_STAGES;
        float v[2][16];
#pragma unroll
        for (int op = 0; op < 2; ++op)
#pragma unroll
            for (int j = 0; j < 16; ++j) v[op][j] = ring[slot][op][j][tid];
        publish(v);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // B: operands of step t published
        multiply();
    }
    if (tail) {
        float v[2][16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int64_t lrow = n0 + 16 * (int64_t)full + j;
            const bool ok = lrow < n1;
            const int64_t row = ok ? rows_at(rw, lrow) : 0;
            v[0][j] = ok ? dY[row * 256 + tid] : 0.0f;
            v[1][j] = ok ? X[row * 256 + tid] : 0.0f;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // operands of the last full step fetched
        publish(v);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        multiply();
    }
    const float back = sy * sx;
#pragma unroll
    for (int a = 0; a < OT; ++a)
#pragma unroll
        for (int b = 0; b < IT; ++b) {
            const int j = (wi * IT + b) * 32 + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = (wo * OT + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                atomicAdd(dW + (int64_t)i * ldw + j, acc[a][b][r] * back);
            }
        }
    if (dbias) atomicAdd(dbias + tid, bsum);
}

// Round 4: the same product, SOFTWARE-PIPELINED.  In k_t_wgrad16c a step runs in phases - wait for the DMA, barrier, read the ring and
// split (VALU + LDS), barrier, 48 MFMAs - and a wave issues in order: while it is feeding its 48 MFMAs into the matrix pipe (~1500
// cycles) it converts nothing, and while it converts the pipe idles (MFMA busy 24 %, a step takes 2.0-2.7 us against 0.7 us of MFMA
// issue).  Here the operands live in TWO buffers: the conversion of step t + 1 (ring -> split halves -> operand buffer (t + 1) & 1) is
// written BETWEEN the MFMAs of step t in program order (sched_group_barrier: one MFMA, then a handful of VALU / LDS instructions),
// the wait for the DMA of step t + 2 moves to the end of the step, and ONE barrier per step certifies all three hand-overs (operands
// of t + 1 published, ring slot t + 1 consumed, step t + 2 landed).  LDS: 3 ring stages (96 KB) + 2 operand buffers (64 KB) = the same
// 160 KB.  Same products in the same order per accumulator: the same partial sums as k_t_wgrad16c (DSN_WGRAD16=c keeps that kernel).
#define W16D_STAGES 3
__global__ void __launch_bounds__(256) k_t_wgrad16d(const float* __restrict__ dY, const float* __restrict__ sy_ptr,
                                                     const float* __restrict__ X, const float* __restrict__ sx_ptr, int64_t N,
                                                     int rows_per_wg, float* __restrict__ dW, int ldw,
                                                     float* __restrict__ dbias, Rows rw, float* __restrict__ part, WgradOps ops) {
    DSN_OWN_SIMD_T(4);
    constexpr int OT = 4, IT = 4, WI = 2;
    __shared__ __attribute__((aligned(16))) float ring[W16D_STAGES][2][16][256];
    __shared__ __attribute__((aligned(16))) t_half8 opbuf[2][2][2][2][256];      // [buffer][operand][hi | lo][8-sample group][feature]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wo = wave / WI, wi = wave % WI;
    const int col = lane & 31, half = lane >> 5;
    const int64_t NL = rows_n(rw, N);
    if (rw.cnt) rows_per_wg = (int)rows_share(NL, 16, 64);     // (the launch was sized for N rows: share what is listed)
    const int64_t n0 = (int64_t)blockIdx.x * rows_per_wg;
    int64_t n1 = n0 + rows_per_wg;
    if (n1 > NL) n1 = NL;
    if (n0 >= n1) return;                                // workgroup-uniform
    const int full = n1 > n0 ? (int)((n1 - n0) >> 4) : 0;
    const bool tail = n0 + 16 * (int64_t)full < n1;
    float bsum = 0.0f;                                   // column sum of dY feature tid (bias gradient)
    // TWO operand pairs into the same accumulators (dY2 != NULL; round 5): the pairs run one after the other over this workgroup's rows;
    // between them the accumulators change units by the ratio of the pairs' scales - powers of two, so exactly.  One partial tile,
    // one reduction for both products of a layer.  The bias gradient (column sums of dY) belongs to the LAST pair.
    const float* jY = dY;
    const float* jX = X;
    float sy = sy_ptr ? t_pow2_at_least(*sy_ptr) : 1.0f, sx = sx_ptr ? t_pow2_at_least(*sx_ptr) : 1.0f;
    float iy = 1.0f / sy, ix = 1.0f / sx;
    bool bias_job = ops.jobs == 1;
    t_f32x16 acc[OT][IT];
#pragma unroll
    for (int a = 0; a < OT; ++a)
#pragma unroll
        for (int b = 0; b < IT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    const unsigned ring_off = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)&ring[0][0][0][0];
    t_i32x4 ids = {0, 0, 0, 0};                          // listed rows: the wave's four row numbers of the next step to be staged
    auto fetch_ids = [&](int t) {
        if (!rw.list) return;
        ids = *reinterpret_cast<t_cptr4>((uintptr_t)(rw.list + (n0 + 16 * (int64_t)t + 4 * wave)));
    };
    // DMA of step t into ring slot t % 3: exactly 8 instructions per wave in every form (the vmcnt waits below count on it)
    auto stage = [&](int t) {
        const int slot = t % W16D_STAGES;
        const int64_t row = n0 + 16 * (int64_t)t + 4 * wave;
        const unsigned da = ring_off + (unsigned)(((slot * 2 + 0) * 16 + 4 * wave) * 1024);
        const unsigned db = ring_off + (unsigned)(((slot * 2 + 1) * 16 + 4 * wave) * 1024);
        int64_t r0 = row;
        if (rw.list) {
            if (ids[3] - ids[0] != 3) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const char* ga = reinterpret_cast<const char*>(jY + (int64_t)ids[j] * 256) + lane * 16;
                    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(ga), "s"(da + 1024u * j) : "memory", "m0");
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const char* gb = reinterpret_cast<const char*>(jX + (int64_t)ids[j] * 256) + lane * 16;
                    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gb), "s"(db + 1024u * j) : "memory", "m0");
                }
                return;
            }
            r0 = (int64_t)ids[0];                        // a run of four consecutive rows: the one-address form
        }
        const char* ga = reinterpret_cast<const char*>(jY + r0 * 256) + lane * 16;
        const char* gb = reinterpret_cast<const char*>(jX + r0 * 256) + lane * 16;
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %0, off\n\tglobal_load_lds_dwordx4 %0, off offset:1024\n\t"
                     "global_load_lds_dwordx4 %0, off offset:2048\n\tglobal_load_lds_dwordx4 %0, off offset:3072"
                     : : "v"(ga), "s"(da) : "memory", "m0");
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %0, off\n\tglobal_load_lds_dwordx4 %0, off offset:1024\n\t"
                     "global_load_lds_dwordx4 %0, off offset:2048\n\tglobal_load_lds_dwordx4 %0, off offset:3072"
                     : : "v"(gb), "s"(db) : "memory", "m0");
    };
    auto ring_read = [&](int slot, float (&v)[2][16]) {
#pragma unroll
        for (int op = 0; op < 2; ++op)
#pragma unroll
            for (int j = 0; j < 16; ++j) v[op][j] = ring[slot][op][j][tid];
    };
    // split the 16 values v[op][row] of feature tid and publish them in operand buffer `buf`
    auto publish = [&](const float (&v)[2][16], int buf) {
#pragma unroll
        for (int j = 0; j < 16; ++j) bsum += bias_job ? v[0][j] : 0.0f;
#pragma unroll
        for (int op = 0; op < 2; ++op) {
            const float inv = op == 0 ? iy : ix;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                t_half8 hi, lo;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float x = v[op][8 * g + j] * inv;
                    const _Float16 h = (_Float16)x;
                    hi[j] = h;
                    lo[j] = (_Float16)(x - (float)h);
                }
                opbuf[buf][op][0][g][tid] = hi;
                opbuf[buf][op][1][g][tid] = lo;
            }
        }
    };
    auto operands = [&](int buf, t_half8 (&ah)[OT], t_half8 (&al)[OT], t_half8 (&bh)[IT], t_half8 (&bl)[IT]) {
#pragma unroll
        for (int a = 0; a < OT; ++a) {
            ah[a] = opbuf[buf][0][0][half][(wo * OT + a) * 32 + col];
            al[a] = opbuf[buf][0][1][half][(wo * OT + a) * 32 + col];
        }
#pragma unroll
        for (int b = 0; b < IT; ++b) {
            bh[b] = opbuf[buf][1][0][half][(wi * IT + b) * 32 + col];
            bl[b] = opbuf[buf][1][1][half][(wi * IT + b) * 32 + col];
        }
    };
    auto mfmas = [&](const t_half8 (&ah)[OT], const t_half8 (&al)[OT], const t_half8 (&bh)[IT], const t_half8 (&bl)[IT]) {
#pragma unroll
        for (int a = 0; a < OT; ++a)
#pragma unroll
            for (int b = 0; b < IT; ++b) {
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bh[b], acc[a][b], 0, 0, 0);
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bl[b], acc[a][b], 0, 0, 0);
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[a], bh[b], acc[a][b], 0, 0, 0);
            }
    };
#pragma nounroll
    for (int job = 0; job < ops.jobs; ++job) {
    if (job == 1) {
        // the second pair: same rows, its own operands and scales; the accumulators go from units of (sy sx) to units of (sy2 sx2)
        const float* const sy2_ptr = ops.sy2;
        const float* const sx2_ptr = ops.sx2;
        const float sy2 = sy2_ptr ? t_pow2_at_least(*sy2_ptr) : 1.0f, sx2 = sx2_ptr ? t_pow2_at_least(*sx2_ptr) : 1.0f;
        const float ratio = t_pair_ratio(sy * sx, sy2 * sx2);
#pragma unroll
        for (int a = 0; a < OT; ++a)
#pragma unroll
            for (int b = 0; b < IT; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] *= ratio;
        jY = ops.dY2; jX = ops.X2;
        sy = sy2; sx = sx2;
        iy = 1.0f / sy; ix = 1.0f / sx;
        bias_job = true;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // every wave has fetched the first pair's last operands
    }
    if (full > 0) {
        const int pre = full < W16D_STAGES ? full : W16D_STAGES;
        for (int t = 0; t < pre; ++t) { fetch_ids(t); stage(t); }
        if (pre < full) fetch_ids(pre);
        // step 0 has landed (steps 1, 2 may fly) -> its operands into buffer 0; then step 1 has landed too
        if (pre == 3) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (pre == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        {
            float v[2][16];
            ring_read(0, v);
            publish(v, 0);
        }
        if (pre == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        for (int t = 0; t + 1 < full; ++t) {
            // here: operands of step t published in buffer t & 1, step t + 1 landed (slot (t + 1) % 3), step t + 2 in flight, slot t % 3 free
            if (t + W16D_STAGES < full) {
                stage(t + W16D_STAGES);
                if (t + W16D_STAGES + 1 < full) fetch_ids(t + W16D_STAGES + 1);
            }
            {
                t_half8 ah[OT], al[OT], bh[IT], bl[IT];
                float v[2][16];
                operands(t & 1, ah, al, bh, bl);
                ring_read((t + 1) % W16D_STAGES, v);
                mfmas(ah, al, bh, bl);
                publish(v, (t + 1) & 1);
#if !defined(W16D_NO_SGB)
                // program order for the in-order issue: operand + ring reads first, then one MFMA : a few VALU, the LDS writes behind
#pragma unroll
                for (int i = 0; i < 48; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, W16D_VALU, 0);
                    if ((i % 6) == 5) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                }
#endif
            }
            // step t + 2 has landed (step t + 3, just issued, may fly)
            if (t + 2 < full) {
                if (t + W16D_STAGES < full) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        {   // the last full step: nothing behind it to convert
            t_half8 ah[OT], al[OT], bh[IT], bl[IT];
            operands((full - 1) & 1, ah, al, bh, bl);
            mfmas(ah, al, bh, bl);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }
    if (tail) {
        float v[2][16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int64_t lrow = n0 + 16 * (int64_t)full + j;
            const bool ok = lrow < n1;
            const int64_t row = ok ? rows_at(rw, lrow) : 0;
            v[0][j] = ok ? jY[row * 256 + tid] : 0.0f;
            v[1][j] = ok ? jX[row * 256 + tid] : 0.0f;
        }
        publish(v, full & 1);                                                   // (that buffer's readers are behind the loop's last barrier)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        t_half8 ah[OT], al[OT], bh[IT], bl[IT];
        operands(full & 1, ah, al, bh, bl);
        mfmas(ah, al, bh, bl);
    }
    }       // pairs
    const float back = sy * sx;
    if (part) {
        // this workgroup's partial tile (unscaled) + its column sums, plain coalesced stores: k_t_wgrad_reduce adds the workgroups up in
        // a fixed order (the 65 536 float atomics per workgroup this replaces cost 0.087 of the kernel's 0.22 ms, and their order - hence
        // the rounding of every gradient - changed from run to run)
        // (in the accumulators' own order - piece (wave, a, b, r / 4) = 64 lanes x 4 registers, one 1 KB store per wave-instruction;
        //  k_t_wgrad_reduce knows where the four values of a lane belong)
        float4* const mine = reinterpret_cast<float4*>(part + (size_t)blockIdx.x * W16_PART(256));
#pragma unroll
        for (int a = 0; a < OT; ++a)
#pragma unroll
            for (int b = 0; b < IT; ++b)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4)
                    mine[(((wave * OT + a) * IT + b) * 4 + r4) * 64 + lane] =
                        make_float4(acc[a][b][4 * r4], acc[a][b][4 * r4 + 1], acc[a][b][4 * r4 + 2], acc[a][b][4 * r4 + 3]);
        part[(size_t)blockIdx.x * W16_PART(256) + 256 * 256 + tid] = bsum;
        return;
    }
#pragma unroll
    for (int a = 0; a < OT; ++a)
#pragma unroll
        for (int b = 0; b < IT; ++b) {
            const int j = (wi * IT + b) * 32 + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = (wo * OT + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                atomicAdd(dW + (int64_t)i * ldw + j, acc[a][b][r] * back);
            }
        }
    if (dbias) atomicAdd(dbias + tid, bsum);
}

// Second stage of the split-fp16 weight-gradient products: dW[i][j] += back * sum over the workgroups g of part[g][i][j], in ascending g -
// a fixed order, so the gradients are bit-reproducible from run to run (the atomicAdd epilogue's were not).  `groups` workgroups were
// launched; with a row list only those whose share of the listed rows is not empty have written (the same share rule as stage one).
// C = columns of the tile (256, or 64 of which `in_valid` are written).  A workgroup owns one 1 KB piece of the partials (64 lanes x
// 4 accumulator registers, in stage one's own order); its four waves take every fourth partial each and meet in LDS.  Workgroups
// 0 .. 3 also add up 64 bias columns each.
template <int C>
__global__ void __launch_bounds__(256) k_t_wgrad_reduce(const float* __restrict__ part, int64_t N, int groups, int rows_per_wg, Rows rw,
                                                         const float* __restrict__ sy_ptr, const float* __restrict__ sx_ptr,
                                                         float* __restrict__ dW, int ldw, int in_valid, float* __restrict__ dbias) {
    __shared__ float4 s_sum[4][64];
    __shared__ float s_b[4][64];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int64_t NL = rows_n(rw, N);
    const int64_t rows = rw.cnt ? rows_share_of(NL, groups, 16, 64) : (int64_t)rows_per_wg;      // stage one's share rule
    const int active = (int)((NL + rows - 1) / rows);
    const int o = blockIdx.x * 64 + lane;            // float4 piece of the tile
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* src = part + (size_t)o * 4;
    // (the tiles are added in the same fixed order as ever; RED_U of them are requested together - the loop is bound by the latency of
    //  its loads, and the four bias blocks' second loop, one load at a time until the end of round 6, was the whole kernel's tail)
    constexpr int RED_U = 8;
    int g = q;
    for (; g + 4 * (RED_U - 1) < active; g += 4 * RED_U) {
        float4 a[RED_U];
#pragma unroll
        for (int u = 0; u < RED_U; ++u) a[u] = *reinterpret_cast<const float4*>(src + (size_t)(g + 4 * u) * W16_PART(C));
#pragma unroll
        for (int u = 0; u < RED_U; ++u) { acc.x += a[u].x; acc.y += a[u].y; acc.z += a[u].z; acc.w += a[u].w; }
    }
    for (; g < active; g += 4) {
        const float4 a0 = *reinterpret_cast<const float4*>(src + (size_t)g * W16_PART(C));
        acc.x += a0.x; acc.y += a0.y; acc.z += a0.z; acc.w += a0.w;
    }
    s_sum[q][lane] = acc;
    float bs = 0.0f;
    const bool bias_block = dbias && blockIdx.x < 4;             // workgroup-uniform
    if (bias_block) {
        const float* bsrc = part + 256 * C + blockIdx.x * 64 + lane;
        int gg = q;
        for (; gg + 4 * (RED_U - 1) < active; gg += 4 * RED_U) {
            float b[RED_U];
#pragma unroll
            for (int u = 0; u < RED_U; ++u) b[u] = bsrc[(size_t)(gg + 4 * u) * W16_PART(C)];
#pragma unroll
            for (int u = 0; u < RED_U; ++u) bs += b[u];
        }
        for (; gg < active; gg += 4) bs += bsrc[(size_t)gg * W16_PART(C)];
        s_b[q][lane] = bs;
    }
    __syncthreads();
    if (q != 0) return;
    const float back = (sy_ptr ? t_pow2_at_least(*sy_ptr) : 1.0f) * (sx_ptr ? t_pow2_at_least(*sx_ptr) : 1.0f);
    float4 t = s_sum[0][lane];
    const float4 t1 = s_sum[1][lane], t2 = s_sum[2][lane], t3 = s_sum[3][lane];
    t.x = ((t.x + t1.x) + t2.x) + t3.x; t.y = ((t.y + t1.y) + t2.y) + t3.y;
    t.z = ((t.z + t1.z) + t2.z) + t3.z; t.w = ((t.w + t1.w) + t2.w) + t3.w;
    // piece blockIdx.x = ((wave * OT + a) * IT + b) * 4 + r4 of the stage-one kernel's accumulators (k_t_wgrad16d: OT = IT = 4, waves 2 x 2;
    // k_t_wgrad16p: 2 x 2 tiles per wave, wave = 64 output rows): the lane's four values are rows i .. i + 3 of column j
    const int r4 = blockIdx.x & 3, half = lane >> 5, col = lane & 31;
    int i, j;
    if (C == 256) {
        const int b = (blockIdx.x >> 2) & 3, a = (blockIdx.x >> 4) & 3, wv = blockIdx.x >> 6;
        i = ((wv >> 1) * 4 + a) * 32 + 8 * r4 + 4 * half;
        j = ((wv & 1) * 4 + b) * 32 + col;
    } else {
        const int b = (blockIdx.x >> 2) & 1, a = (blockIdx.x >> 3) & 1, wv = blockIdx.x >> 4;
        i = 64 * wv + 32 * a + 8 * r4 + 4 * half;
        j = 32 * b + col;
    }
    if (j < in_valid) {
        float* out = dW + (int64_t)i * ldw + j;
        out[0] += t.x * back;
        out[(int64_t)ldw] += t.y * back;
        out[2 * (int64_t)ldw] += t.z * back;
        out[3 * (int64_t)ldw] += t.w * back;
    }
    if (bias_block) dbias[blockIdx.x * 64 + lane] += ((s_b[0][lane] + s_b[1][lane]) + s_b[2][lane]) + s_b[3][lane];
}

// The 256 x 64 sibling for the positional-encoding columns: dW[256, 63] += dY[N,256]^T X[N,64] (X = pe or its tangent).  Same
// machinery with smaller pieces - 3 stages of (16 KB + 4 KB), 20 KB of operands, 80 KB of LDS = two workgroups per CU; wave w
// owns output rows 64 w .. 64 w + 63 and all 64 columns (2 x 2 tiles).  Column 63 is the zero pad and is not written.
#define W16P_STAGES 3
__global__ void __launch_bounds__(256, 2) k_t_wgrad16p(const float* __restrict__ dY, const float* __restrict__ sy_ptr,
                                                        const float* __restrict__ X, const float* __restrict__ sx_ptr, int64_t N,
                                                        int rows_per_wg, float* __restrict__ dW, int ldw, int in_valid,
                                                        float* __restrict__ dbias, Rows rw, float* __restrict__ part, WgradOps ops) {
    // (no DSN_OWN_SIMD: K = 8 MFMAs only - see multiply())
    __shared__ __attribute__((aligned(16))) float ringY[W16P_STAGES][16][256];
    __shared__ __attribute__((aligned(16))) float ringX[W16P_STAGES][16][64];
    __shared__ __attribute__((aligned(16))) t_half8 opY[2][2][256];
    __shared__ __attribute__((aligned(16))) t_half8 opX[2][2][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, half = lane >> 5;
    const int64_t NL = rows_n(rw, N);
    if (rw.cnt) rows_per_wg = (int)rows_share(NL, 16, 64);
    const int64_t n0 = (int64_t)blockIdx.x * rows_per_wg;
    int64_t n1 = n0 + rows_per_wg;
    if (n1 > NL) n1 = NL;
    if (n0 >= n1) return;                                // workgroup-uniform
    const int full = n1 > n0 ? (int)((n1 - n0) >> 4) : 0;
    const bool tail = n0 + 16 * (int64_t)full < n1;
    float bsum = 0.0f;
    // (two operand pairs into the same accumulators when dY2 != NULL: see k_t_wgrad16d)
    const float* jY = dY;
    const float* jX = X;
    float sy = sy_ptr ? t_pow2_at_least(*sy_ptr) : 1.0f, sx = sx_ptr ? t_pow2_at_least(*sx_ptr) : 1.0f;
    float iy = 1.0f / sy, ix = 1.0f / sx;
    bool bias_job = ops.jobs == 1;
    t_f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    const unsigned offY = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)&ringY[0][0][0];
    const unsigned offX = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)&ringX[0][0][0];
    t_i32x4 ids = {0, 0, 0, 0};        // listed rows: this wave's four row numbers of the NEXT step (one scalar load, see k_t_wgrad16c)
    auto fetch_ids = [&](int t) {
        if (!rw.list) return;
        ids = *reinterpret_cast<t_cptr4>((uintptr_t)(rw.list + (n0 + 16 * (int64_t)t + 4 * wave)));
    };
    // DMA of step t: rows 4 wave .. + 3 of dY (1 KB each) and KB `wave` of the 4 KB the 16 rows of X occupy
    auto stage = [&](int t) {
        const int slot = t % W16P_STAGES;
        const int64_t row = n0 + 16 * (int64_t)t;
        const unsigned da = offY + (unsigned)((slot * 16 + 4 * wave) * 1024);
        const unsigned db = offX + (unsigned)(slot * 4096 + wave * 1024);
        if (rw.list) {
            // listed rows: one address per 1 KB row of dY; the wave's KB of X is four 256-byte rows, lane -> (row lane / 16, 16 B piece).
            // Row numbers fetched one step earlier (scalar loads; a vector load's wait would drain the DMA queue)
            const int rr[4] = {ids[0], ids[1], ids[2], ids[3]};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const char* ga = reinterpret_cast<const char*>(jY + (int64_t)rr[j] * 256) + lane * 16;
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(ga), "s"(da + 1024u * j) : "memory", "m0");
            }
            const int q = lane >> 4;
            const int64_t rx = q == 0 ? rr[0] : (q == 1 ? rr[1] : (q == 2 ? rr[2] : rr[3]));
            const char* gb = reinterpret_cast<const char*>(jX + rx * 64) + (lane & 15) * 16;
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gb), "s"(db) : "memory", "m0");
            return;
        }
        const char* ga = reinterpret_cast<const char*>(jY + (row + 4 * wave) * 256) + lane * 16;
        const char* gb = reinterpret_cast<const char*>(jX + row * 64) + wave * 1024 + lane * 16;
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %0, off\n\tglobal_load_lds_dwordx4 %0, off offset:1024\n\t"
                     "global_load_lds_dwordx4 %0, off offset:2048\n\tglobal_load_lds_dwordx4 %0, off offset:3072"
                     : : "v"(ga), "s"(da) : "memory", "m0");
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gb), "s"(db) : "memory", "m0");
    };
    auto split8 = [&](const float* v, float inv, t_half8& hi, t_half8& lo) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = v[j] * inv;
            const _Float16 h = (_Float16)x;
            hi[j] = h;
            lo[j] = (_Float16)(x - (float)h);
        }
    };
    // thread f splits feature f of dY (16 rows); threads 0..127 split feature f & 63 of X for the 8-row group f >> 6
    auto publish = [&](const float (&vy)[16], const float (&vx)[8]) {
#pragma unroll
        for (int j = 0; j < 16; ++j) bsum += bias_job ? vy[j] : 0.0f;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            t_half8 hi, lo;
            split8(vy + 8 * g, iy, hi, lo);
            opY[0][g][tid] = hi;
            opY[1][g][tid] = lo;
        }
        if (tid < 128) {
            t_half8 hi, lo;
            split8(vx, ix, hi, lo);
            opX[0][tid >> 6][tid & 63] = hi;
            opX[1][tid >> 6][tid & 63] = lo;
        }
    };
    // Round 6: the products run on v_mfma_f32_32x32x8_f16 (K = 8, the CDNA3-era instruction), two per 16-row step and operand pair,
    // not on gfx950's K = 16 form.  Waves that issue v_mfma_f32_32x32x16_f16 make co-resident waves of OTHER kernels consume their
    // vector-memory loads early (dsn_common.h DSN_OWN_SIMD; found by exchanging exactly this instruction, profiles/r06_coresidency_bisect.txt);
    // such kernels must own their SIMDs.  This one is bound by its operand stream (0.75 GB in 0.145 ms), its matrix work is a quarter of
    // its time even at the K = 8 rate - so it keeps two workgroups per CU and needs no guard.  A K = 8 operand is half of the 8-sample
    // vector the conversion stage publishes: lanes 0-31 take samples 0-3 of the group, lanes 32-63 samples 4-7.
    typedef _Float16 t_half4 __attribute__((ext_vector_type(4)));
    auto multiply = [&]() {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            t_half4 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                ah[a] = reinterpret_cast<const t_half4*>(&opY[0][g][64 * wave + 32 * a + col])[half];
                al[a] = reinterpret_cast<const t_half4*>(&opY[1][g][64 * wave + 32 * a + col])[half];
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                bh[b] = reinterpret_cast<const t_half4*>(&opX[0][g][32 * b + col])[half];
                bl[b] = reinterpret_cast<const t_half4*>(&opX[1][g][32 * b + col])[half];
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x8f16(ah[a], bh[b], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x8f16(ah[a], bl[b], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x8f16(al[a], bh[b], acc[a][b], 0, 0, 0);
                }
        }
    };
#pragma nounroll
    for (int job = 0; job < ops.jobs; ++job) {
    if (job == 1) {
        const float* const sy2_ptr = ops.sy2;
        const float* const sx2_ptr = ops.sx2;
        const float sy2 = sy2_ptr ? t_pow2_at_least(*sy2_ptr) : 1.0f, sx2 = sx2_ptr ? t_pow2_at_least(*sx2_ptr) : 1.0f;
        const float ratio = t_pair_ratio(sy * sx, sy2 * sx2);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] *= ratio;
        jY = ops.dY2; jX = ops.X2;
        sy = sy2; sx = sx2;
        iy = 1.0f / sy; ix = 1.0f / sx;
        bias_job = true;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    const int pre = full < W16P_STAGES - 1 ? full : W16P_STAGES - 1;
    for (int t = 0; t < pre; ++t) { fetch_ids(t); stage(t); }
    if (pre < full) fetch_ids(pre);
    for (int t = 0; t < full; ++t) {
        const int issued = (t + W16P_STAGES - 1 < full) ? t + W16P_STAGES - 1 : full;
        if (issued - (t + 1) >= 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");       // one later step (5 pieces) may still fly
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (t + W16P_STAGES - 1 < full) {
            stage(t + W16P_STAGES - 1);
            if (t + W16P_STAGES < full) fetch_ids(t + W16P_STAGES);
        }
        const int slot = t % W16P_STAGES;
        float vy[16], vx[8];
#pragma unroll
        for (int j = 0; j < 16; ++j) vy[j] = ringY[slot][j][tid];
#pragma unroll
        for (int j = 0; j < 8; ++j) vx[j] = ringX[slot][8 * ((tid >> 6) & 1) + j][tid & 63];
        publish(vy, vx);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        multiply();
    }
    if (tail) {
        float vy[16], vx[8];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int64_t lrow = n0 + 16 * (int64_t)full + j;
            vy[j] = lrow < n1 ? jY[rows_at(rw, lrow) * 256 + tid] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t lrow = n0 + 16 * (int64_t)full + 8 * ((tid >> 6) & 1) + j;
            vx[j] = lrow < n1 ? jX[rows_at(rw, lrow) * 64 + (tid & 63)] : 0.0f;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        publish(vy, vx);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        multiply();
    }
    }       // pairs
    const float back = sy * sx;
    if (part) {      // two-stage form: see k_t_wgrad16d / k_t_wgrad_reduce
        float4* const mine = reinterpret_cast<float4*>(part + (size_t)blockIdx.x * W16_PART(64));
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4)
                    mine[(((wave * 2 + a) * 2 + b) * 4 + r4) * 64 + lane] =
                        make_float4(acc[a][b][4 * r4], acc[a][b][4 * r4 + 1], acc[a][b][4 * r4 + 2], acc[a][b][4 * r4 + 3]);
        part[(size_t)blockIdx.x * W16_PART(64) + 256 * 64 + tid] = bsum;
        return;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int j = 32 * b + col;
            if (j >= in_valid) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = 64 * wave + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * half;
                atomicAdd(dW + (int64_t)i * ldw + j, acc[a][b][r] * back);
            }
        }
    if (dbias) atomicAdd(dbias + tid, bsum);
}

// The 128-row siblings for the lighting MLP and the colour head (round 6): dW[128, XW] += dY[N,128]^T X[N,XW] with
// XW = 256 (rgb_net.1: dY = d_rr, X = h_6) or 128 (lights_encoding.2: dY = d_hl2, X = hl1), + the bias gradient (column sums of dY).
// These ran on the exact-fp32 MFMA kernel k_t_wgrad (0.32 + 0.17 ms per 8192 x 64 step at 57 TFLOP/s): the contraction over the
// batch is the same as in the trunk, and so is the remedy - the split-fp16 machinery of k_t_wgrad16p with other tile shapes: wave w
// owns output rows 32 w .. 32 w + 31 and all XW columns (1 x XW / 32 tiles); a step stages 16 rows of both operands through LDS
// (2 + XW / 128 DMA instructions per wave), thread f splits feature f & 127 of dY for the 8-row group f >> 7 and feature(s) of X.
// Two-stage only: every workgroup leaves its [128][XW] tile + 128 column sums in `part`, k_t_wgrad_reduce_q adds them in a fixed order.
#define W16Q_STAGES 3
template <int XW>
__global__ void __launch_bounds__(256, 1) k_t_wgrad16q(const float* __restrict__ dY, const float* __restrict__ sy_ptr,
                                                        const float* __restrict__ X, const float* __restrict__ sx_ptr, int64_t N,
                                                        int rows_per_wg, Rows rw, float* __restrict__ part) {
    DSN_OWN_SIMD_T(128);
    constexpr int YW = 128, XB = XW / 32, XI = XW / 128;      // XI: DMA instructions of X per wave and step (1 KB = 256 / XW rows each)
    constexpr int PIECES = 2 + XI;                             // DMA instructions per wave and step
    constexpr int PART = YW * XW + 256;
    __shared__ __attribute__((aligned(16))) float ringY[W16Q_STAGES][16][YW];
    __shared__ __attribute__((aligned(16))) float ringX[W16Q_STAGES][16][XW];
    __shared__ __attribute__((aligned(16))) t_half8 opY[2][2][YW];
    __shared__ __attribute__((aligned(16))) t_half8 opX[2][2][XW];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, half = lane >> 5;
    const int64_t NL = rows_n(rw, N);
    if (rw.cnt) rows_per_wg = (int)rows_share(NL, 16, 64);
    const int64_t n0 = (int64_t)blockIdx.x * rows_per_wg;
    int64_t n1 = n0 + rows_per_wg;
    if (n1 > NL) n1 = NL;
    if (n0 >= n1) return;                                // workgroup-uniform
    const int full = (int)((n1 - n0) >> 4);
    const bool tail = n0 + 16 * (int64_t)full < n1;
    float bsum = 0.0f;
    const float sy = sy_ptr ? t_pow2_at_least(*sy_ptr) : 1.0f, sx = sx_ptr ? t_pow2_at_least(*sx_ptr) : 1.0f;
    const float iy = 1.0f / sy, ix = 1.0f / sx;
    t_f32x16 acc[XB];
#pragma unroll
    for (int b = 0; b < XB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.0f;
    const unsigned offY = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)&ringY[0][0][0];
    const unsigned offX = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)&ringX[0][0][0];
    t_i32x4 ids = {0, 0, 0, 0};        // listed rows: this wave's four row numbers of the NEXT step (one scalar load, see k_t_wgrad16c)
    auto fetch_ids = [&](int t) {
        if (!rw.list) return;
        ids = *reinterpret_cast<t_cptr4>((uintptr_t)(rw.list + (n0 + 16 * (int64_t)t + 4 * wave)));
    };
    // DMA of step t: the wave's rows 4 wave .. + 3 of both operands; a 1 KB instruction covers 2 rows of a 128-wide operand (lane ->
    // row lane / 32, 16-byte piece lane % 32) and 1 row of a 256-wide one
    auto stage = [&](int t) {
        const int slot = t % W16Q_STAGES;
        const int64_t row = n0 + 16 * (int64_t)t + 4 * wave;
        const unsigned da = offY + (unsigned)((slot * 16 + 4 * wave) * (YW * 4));
        const unsigned db = offX + (unsigned)((slot * 16 + 4 * wave) * (XW * 4));
        const int hr = lane >> 5, pc = lane & 31;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t r = rw.list ? (int64_t)(hr ? ids[2 * j + 1] : ids[2 * j]) : row + 2 * j + hr;
            const char* ga = reinterpret_cast<const char*>(dY + r * YW) + pc * 16;
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(ga), "s"(da + 1024u * j) : "memory", "m0");
        }
        if (XW == 128) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int64_t r = rw.list ? (int64_t)(hr ? ids[2 * j + 1] : ids[2 * j]) : row + 2 * j + hr;
                const char* gb = reinterpret_cast<const char*>(X + r * XW) + pc * 16;
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gb), "s"(db + 1024u * j) : "memory", "m0");
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t r = rw.list ? (int64_t)ids[j] : row + j;
                const char* gb = reinterpret_cast<const char*>(X + r * XW) + lane * 16;
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gb), "s"(db + 1024u * j) : "memory", "m0");
            }
        }
    };
    auto split8 = [&](const float* v, float inv, t_half8& hi, t_half8& lo) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = v[j] * inv;
            const _Float16 h = (_Float16)x;
            hi[j] = h;
            lo[j] = (_Float16)(x - (float)h);
        }
    };
    constexpr int XG = XW == 128 ? 1 : 2;      // 8-row groups of X a thread splits (XW = 256: thread f takes feature f, both groups)
    // thread f splits feature f & 127 of dY for the 8-row group f >> 7; of X: feature f & 127, group f >> 7 (XW = 128) or feature f, both groups
    auto publish = [&](const float (&vy)[8], const float (&vx)[8 * XG]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) bsum += vy[j];
        {
            t_half8 hi, lo;
            split8(vy, iy, hi, lo);
            opY[0][tid >> 7][tid & 127] = hi;
            opY[1][tid >> 7][tid & 127] = lo;
        }
#pragma unroll
        for (int g = 0; g < XG; ++g) {
            t_half8 hi, lo;
            split8(vx + 8 * g, ix, hi, lo);
            if (XW == 128) { opX[0][tid >> 7][tid & 127] = hi; opX[1][tid >> 7][tid & 127] = lo; }
            else { opX[0][g][tid] = hi; opX[1][g][tid] = lo; }
        }
    };
    // W16Q_ACC (experiment builds): where the accumulators live - 0: the compiler's choice (AGPRs here), 1: AGPRs by constraint,
    // 2: architectural VGPRs by constraint.  (Round 6, co-residency hazard of DESIGN 4.5: is "f16 MFMA into AGPR accumulators" the aggressor?)
#ifndef W16Q_ACC
#define W16Q_ACC 0
#endif
    auto mfma = [&](const t_half8& a, const t_half8& b, t_f32x16& c) {
#if defined(DSN_EXPERIMENTS) && W16Q_ACC == 1
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
#elif defined(DSN_EXPERIMENTS) && W16Q_ACC == 2
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
#elif defined(DSN_EXPERIMENTS) && W16Q_ACC == 3
        // (WRONG products, timing / hazard experiments only: the CDNA3-era K = 8 instruction twice instead of gfx950's K = 16 one)
        typedef _Float16 t_half4 __attribute__((ext_vector_type(4)));
        const t_half4 a0 = {a[0], a[1], a[2], a[3]}, a1 = {a[4], a[5], a[6], a[7]}, b0 = {b[0], b[1], b[2], b[3]}, b1 = {b[4], b[5], b[6], b[7]};
        c = __builtin_amdgcn_mfma_f32_32x32x8f16(a0, b0, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x8f16(a1, b1, c, 0, 0, 0);
#else
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#endif
    };
    auto multiply = [&]() {
        const t_half8 ah = opY[0][half][32 * wave + col], al = opY[1][half][32 * wave + col];
#pragma unroll
        for (int b = 0; b < XB; ++b) {
            const t_half8 bh = opX[0][half][32 * b + col], bl = opX[1][half][32 * b + col];
            mfma(ah, bh, acc[b]);
            mfma(ah, bl, acc[b]);
            mfma(al, bh, acc[b]);
        }
    };
    const int pre = full < W16Q_STAGES - 1 ? full : W16Q_STAGES - 1;
    for (int t = 0; t < pre; ++t) { fetch_ids(t); stage(t); }
    if (pre < full) fetch_ids(pre);
    for (int t = 0; t < full; ++t) {
        const int issued = (t + W16Q_STAGES - 1 < full) ? t + W16Q_STAGES - 1 : full;
        if (issued - (t + 1) >= 1) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(PIECES) : "memory");       // one later step may still fly
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (t + W16Q_STAGES - 1 < full) {
            stage(t + W16Q_STAGES - 1);
            if (t + W16Q_STAGES < full) fetch_ids(t + W16Q_STAGES);
        }
        const int slot = t % W16Q_STAGES;
        float vy[8], vx[8 * XG];
#pragma unroll
        for (int j = 0; j < 8; ++j) vy[j] = ringY[slot][8 * (tid >> 7) + j][tid & 127];
        if (XW == 128) {
#pragma unroll
            for (int j = 0; j < 8; ++j) vx[j] = ringX[slot][8 * (tid >> 7) + j][tid & 127];
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) vx[j] = ringX[slot][j][tid & (XW - 1)];
        }
        publish(vy, vx);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        multiply();
    }
    if (tail) {
        float vy[8], vx[8 * XG];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t lrow = n0 + 16 * (int64_t)full + 8 * (tid >> 7) + j;
            vy[j] = lrow < n1 ? dY[rows_at(rw, lrow) * YW + (tid & 127)] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < 8 * XG; ++j) {
            const int64_t lrow = n0 + 16 * (int64_t)full + (XW == 128 ? 8 * (tid >> 7) + j : j);
            vx[j] = lrow < n1 ? X[rows_at(rw, lrow) * XW + (XW == 128 ? (tid & 127) : (tid & (XW - 1)))] : 0.0f;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        publish(vy, vx);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        multiply();
    }
    // the workgroup's tile in the accumulators' own order: piece (wave, b, r / 4) = 64 lanes x 4 registers; then 256 column partial sums
    float4* const mine = reinterpret_cast<float4*>(part + (size_t)blockIdx.x * PART);
#pragma unroll
    for (int b = 0; b < XB; ++b)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4)
            mine[((wave * XB + b) * 4 + r4) * 64 + lane] = make_float4(acc[b][4 * r4], acc[b][4 * r4 + 1], acc[b][4 * r4 + 2], acc[b][4 * r4 + 3]);
    part[(size_t)blockIdx.x * PART + YW * XW + tid] = bsum;       // (feature tid & 127: the two 8-row groups' sums apart)
}
// second stage: dW[i][j] += back * sum_g part[g][i][j] in ascending g (bit-reproducible), dbias[i] += sum_g of both column partial sums
template <int XW>
__global__ void __launch_bounds__(256) k_t_wgrad_reduce_q(const float* __restrict__ part, int64_t N, int groups, int rows_per_wg, Rows rw,
                                                           const float* __restrict__ sy_ptr, const float* __restrict__ sx_ptr,
                                                           float* __restrict__ dW, int ldw, float* __restrict__ dbias) {
    constexpr int YW = 128, XB = XW / 32, PART = YW * XW + 256;
    __shared__ float4 s_sum[4][64];
    __shared__ float s_b[4][64];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int64_t NL = rows_n(rw, N);
    const int64_t rows = rw.cnt ? rows_share_of(NL, groups, 16, 64) : (int64_t)rows_per_wg;      // stage one's share rule
    const int active = (int)((NL + rows - 1) / rows);
    const float* src = part + ((size_t)blockIdx.x * 64 + lane) * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr int RED_U = 8;                                     // (tiles requested together, added in the old order: see k_t_wgrad_reduce)
    int g = q;
    for (; g + 4 * (RED_U - 1) < active; g += 4 * RED_U) {
        float4 a[RED_U];
#pragma unroll
        for (int u = 0; u < RED_U; ++u) a[u] = *reinterpret_cast<const float4*>(src + (size_t)(g + 4 * u) * PART);
#pragma unroll
        for (int u = 0; u < RED_U; ++u) { acc.x += a[u].x; acc.y += a[u].y; acc.z += a[u].z; acc.w += a[u].w; }
    }
    for (; g < active; g += 4) {
        const float4 a0 = *reinterpret_cast<const float4*>(src + (size_t)g * PART);
        acc.x += a0.x; acc.y += a0.y; acc.z += a0.z; acc.w += a0.w;
    }
    s_sum[q][lane] = acc;
    const bool bias_block = dbias && blockIdx.x < 2;             // workgroup-uniform: block b adds up bias columns 64 b .. 64 b + 63
    if (bias_block) {
        float bs = 0.0f;
        const float* bsrc = part + YW * XW + blockIdx.x * 64 + lane;
        int gg = q;
        for (; gg + 4 * (RED_U - 1) < active; gg += 4 * RED_U) {
            float b0[RED_U], b1[RED_U];
#pragma unroll
            for (int u = 0; u < RED_U; ++u) { b0[u] = bsrc[(size_t)(gg + 4 * u) * PART]; b1[u] = bsrc[(size_t)(gg + 4 * u) * PART + 128]; }
#pragma unroll
            for (int u = 0; u < RED_U; ++u) bs += b0[u] + b1[u];
        }
        for (; gg < active; gg += 4) bs += bsrc[(size_t)gg * PART] + bsrc[(size_t)gg * PART + 128];
        s_b[q][lane] = bs;
    }
    __syncthreads();
    if (q != 0) return;
    const float back = (sy_ptr ? t_pow2_at_least(*sy_ptr) : 1.0f) * (sx_ptr ? t_pow2_at_least(*sx_ptr) : 1.0f);
    float4 t = s_sum[0][lane];
    const float4 t1 = s_sum[1][lane], t2 = s_sum[2][lane], t3 = s_sum[3][lane];
    t.x = ((t.x + t1.x) + t2.x) + t3.x; t.y = ((t.y + t1.y) + t2.y) + t3.y;
    t.z = ((t.z + t1.z) + t2.z) + t3.z; t.w = ((t.w + t1.w) + t2.w) + t3.w;
    // piece blockIdx.x = (wave * XB + b) * 4 + r4 of stage one: the lane's four values are rows i .. i + 3 of column j
    const int r4 = blockIdx.x & 3, b = (blockIdx.x >> 2) % XB, wv = (blockIdx.x >> 2) / XB, half = lane >> 5, col = lane & 31;
    const int i = 32 * wv + 8 * r4 + 4 * half, j = 32 * b + col;
    float* out = dW + (int64_t)i * ldw + j;
    out[0] += t.x * back;
    out[(int64_t)ldw] += t.y * back;
    out[2 * (int64_t)ldw] += t.z * back;
    out[3 * (int64_t)ldw] += t.w * back;
    if (bias_block) dbias[blockIdx.x * 64 + lane] += ((s_b[0][lane] + s_b[1][lane]) + s_b[2][lane]) + s_b[3][lane];
}
// dW [128, XW] (ldw) += dY[N,128]^T X[N,XW]; dbias [128] += column sums of dY; sy / sx: device scalars with the operands' batch-wide
// magnitudes (NULL = O(1) operand)
template <int XW>
void wgrad_mfma16q(int64_t N, const float* X, const float* sx, const float* dY, const float* sy, float* dW, int ldw, hipStream_t st,
                   float* dbias, Rows rw, float* part) {
    int groups = 256;                 // one workgroup per CU (48 - 96 KB of LDS)
    int rows = (int)((N + groups - 1) / groups);
    if (rows < 64) rows = 64;
    rows = (rows + 15) & ~15;
    groups = (int)((N + rows - 1) / rows);
    hipLaunchKernelGGL((k_t_wgrad16q<XW>), dim3((unsigned)groups), dim3(256), 0, st, dY, sy, X, sx, N, rows, rw, part);
    hipLaunchKernelGGL((k_t_wgrad_reduce_q<XW>), dim3(128 * XW / 256), dim3(256), 0, st, (const float*)part, N, groups, rows, rw, sy, sx, dW, ldw,
                       dbias);
}

// How the workgroups' partial products meet.  Default: two stages - every workgroup stores its tile into `part` (the training
// workspace's W16_PART_GROUPS x W16_PART(256) floats), k_t_wgrad_reduce adds them in a fixed order: bit-reproducible gradients, and
// 0.13 + 0.02 ms per 256 x 256 product instead of 0.22 (the 65 536 float atomics per workgroup were 40 % of the kernel).
// DSN_WGRAD_REDUCE=atomic: round 1-3's atomicAdd epilogue (A/B switch).
static bool wgrad_two_stage() {
    static const bool v = [] { const char* e = getenv("DSN_WGRAD_REDUCE"); return !(e && e[0] == 'a'); }();
    return v;
}
// A second operand pair (X2, sx2, dY2, sy2) of the same shape adds its product into the same accumulators - ONE launch, one partial
// tile and one reduction for the two weight-gradient products of a layer (tangent pair first, adjoint pair second; the bias gradient
// is the second pair's).  VERDICT r04 #3 (c): 12 + 4 launches -> 6 + 2, half the 64 MB partial bursts and half of k_t_wgrad_reduce.
struct WgradPair { const float* X; const float* sx; const float* dY; const float* sy; };
void wgrad_mfma16p(int64_t N, const float* X, const float* sx, const float* dY, const float* sy, float* dW, int ldw, int in_valid,
                   hipStream_t st, float* dbias = nullptr, Rows rw = Rows{nullptr, nullptr}, float* part = nullptr,
                   WgradPair second = WgradPair{nullptr, nullptr, nullptr, nullptr}) {
    int groups = W16_PART_GROUPS;     // two workgroups per CU
    int rows = (int)((N + groups - 1) / groups);
    if (rows < 64) rows = 64;
    rows = (rows + 15) & ~15;
    groups = (int)((N + rows - 1) / rows);
    if (!wgrad_two_stage()) part = nullptr;
    hipLaunchKernelGGL(k_t_wgrad16p, dim3((unsigned)groups), dim3(256), 0, st, dY, sy, X, sx, N, rows, dW, ldw, in_valid, dbias, rw, part,
                       WgradOps{second.dY, second.sy, second.X, second.sx, second.dY ? 2 : 1});
    const float* const ry = second.dY ? second.sy : sy;      // the partial tiles are in the LAST pair's units
    const float* const rx = second.dY ? second.sx : sx;
    if (part)
        hipLaunchKernelGGL(k_t_wgrad_reduce<64>, dim3(256 * 64 / 256), dim3(256), 0, st, (const float*)part, N, groups, rows, rw, ry, rx, dW, ldw,
                           in_valid, dbias);
}

// dW [256,256] (ldw) += dY[N,256]^T X[N,256], operands scaled by the device scalars sy / sx (NULL = O(1) operand);
// dbias (optional) [256] += column sums of dY
void wgrad_mfma16(int64_t N, const float* X, const float* sx, const float* dY, const float* sy, float* dW, int ldw, hipStream_t st,
                  float* dbias = nullptr, Rows rw = Rows{nullptr, nullptr}, float* part = nullptr,
                  WgradPair second = WgradPair{nullptr, nullptr, nullptr, nullptr}) {
    int groups = 256;                 // one workgroup per CU, one round (with 512 the launch ran 0.33 instead of 0.28 ms)
    int rows = (int)((N + groups - 1) / groups);
    if (rows < 64) rows = 64;
    rows = (rows + 15) & ~15;
    groups = (int)((N + rows - 1) / rows);
    static const bool old_kernel = [] { const char* e = getenv("DSN_WGRAD16"); return e && e[0] == 'c'; }();
    if (old_kernel || !wgrad_two_stage()) part = nullptr;
    if (old_kernel) {      // (round 3's kernel knows one pair: two launches)
        hipLaunchKernelGGL(k_t_wgrad16c, dim3((unsigned)groups), dim3(256), 0, st, dY, sy, X, sx, N, rows, dW, ldw, second.dY ? nullptr : dbias, rw);
        if (second.dY)
            hipLaunchKernelGGL(k_t_wgrad16c, dim3((unsigned)groups), dim3(256), 0, st, second.dY, second.sy, second.X, second.sx, N, rows, dW, ldw,
                               dbias, rw);
        return;
    }
    hipLaunchKernelGGL(k_t_wgrad16d, dim3((unsigned)groups), dim3(256), 0, st, dY, sy, X, sx, N, rows, dW, ldw, dbias, rw, part,
                       WgradOps{second.dY, second.sy, second.X, second.sx, second.dY ? 2 : 1});
    const float* const ry = second.dY ? second.sy : sy;
    const float* const rx = second.dY ? second.sx : sx;
    if (part)
        hipLaunchKernelGGL(k_t_wgrad_reduce<256>, dim3(256 * 256 / 256), dim3(256), 0, st, (const float*)part, N, groups, rows, rw, ry, rx, dW,
                           ldw, 256, dbias);
}


// ------------------------------------------------------------------------------------------------------------
// k_t_lin : Y [N,M] = X [N,K] B  with the small 128 / 256-wide matrices of the lighting MLP and the colour head, exact fp32
// (v_mfma_f32_32x32x2_f32 = an fp32 fma chain), with the element-wise step that follows each of them fused into the
// store.  These three products were the last rocBLAS calls of the library (sgemm + a separate bias / mask / seed pass each).
//   TRANS = false: B = W as stored  ([K,M] row-major: dX = dY W, W = torch Linear weight [out = K, in = M])
//   TRANS = true : B = W^T          (Y = X W^T, W [M,K]: the forward Linear)
// One workgroup per CU: B is staged in LDS once (K M floats, up to 128 KB), then every wave walks 32-row tiles of X: the A
// operand (lane = row, half-wave = k parity group) comes as ONE float4 per 8 k-values - a contraction does not care about the
// order of k, so step s of a group takes k0 + s from the low half-wave and k0 + 4 + s from the high one - and the B operand
// from LDS (lanes = consecutive columns: conflict-free).  The accumulator layout puts 32 consecutive columns of one row in the
// lanes of a half-wave, so every store instruction writes two full 128-byte row segments.
// Epilogues: EPI_NONE; EPI_BIAS_RELU y = relu(y + bias[c]); EPI_MASK y = msrc[n,c] > 0 ? y : 0;
//            EPI_SEED y = msrc[n,c] > 0 ? y + sc[n] wv[c] : 0   (the seed of the adjoint pass, what k_t_seed did on top of the GEMM)
// ------------------------------------------------------------------------------------------------------------
enum { EPI_NONE = 0, EPI_BIAS_RELU = 1, EPI_MASK = 2, EPI_SEED = 3 };
template <int K, int M, bool TRANS, int EPI>
__global__ void __launch_bounds__(256, 1) k_t_lin(const float* __restrict__ X, const float* __restrict__ W, float* __restrict__ Y,
                                                  int64_t N, const float* __restrict__ bias_or_wv, const float* __restrict__ msrc,
                                                  const float* __restrict__ sc, Rows rw, const uint4* __restrict__ recs, int rec_layer) {
    // recs / rec_layer (EPI_SEED, optional; round 6): the relu pattern of trunk layer `rec_layer` from the training forward's 224-byte
    // records - [sample][half][layer] uint4, bit 15 - r of the 16-bit word of block t = accumulator register r (dsn_field16.hip) -
    // instead of `msrc > 0`: 16 bytes per row and half in place of 512 (the pattern the tangent / adjoint kernels use anyway)
    DSN_OWN_SIMD_T(16);
    constexpr int NT = M / 32, NQ = NT / 4;
    // B in LDS as [k][quad of column tiles][column in tile][tile in quad]: the 4 column tiles a lane feeds with one k come
    // back with ONE ds_read_b128 (lanes 16 bytes apart: conflict-free)
    __shared__ __attribute__((aligned(16))) float sB[K * M];
    __shared__ float sV[M];
    const int tid = threadIdx.x;
    for (int i = tid; i < K * M; i += 256) {
        const int k = i / M, m = i % M, t = m >> 5, c = m & 31;
        sB[((k * NQ + (t >> 2)) * 32 + c) * 4 + (t & 3)] = TRANS ? W[m * K + k] : W[i];
    }
    if (EPI == EPI_BIAS_RELU || EPI == EPI_SEED)
        for (int i = tid; i < M; i += 256) sV[i] = bias_or_wv[i];
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;
    const int64_t NL = rows_n(rw, N);
    const int64_t ntile = (NL + 31) / 32;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < ntile; tile += (int64_t)gridDim.x * 4) {
        // D[feature][point] = sum_k B[k][feature] X[point][k]: lane = point `col`, registers = features
        const int64_t lrow = tile * 32 + col;
        const bool valid = lrow < NL;
        const int64_t crow = rows_at(rw, valid ? lrow : NL - 1);
        const int64_t prow = crow;
        const float* xr = X + crow * K + 4 * half;
        // what the epilogue needs from memory is requested NOW and used after the products: its latency hides behind them
        float4 mk[EPI == EPI_MASK || EPI == EPI_SEED ? NT : 1][4];
        float scn = 0.0f;
        const bool by_rec = EPI == EPI_SEED && recs != nullptr;      // kernel-uniform
        uint4 rec = make_uint4(0u, 0u, 0u, 0u);
        if (EPI == EPI_MASK || EPI == EPI_SEED) {
            if (by_rec) rec = recs[((size_t)crow * 2 + half) * 7 + rec_layer];
            else {
                const float* mr = msrc + crow * M + 4 * half;
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) mk[t][q] = *reinterpret_cast<const float4*>(mr + 32 * t + 8 * q);
            }
            if (EPI == EPI_SEED) scn = sc[crow];
        }
        t_f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
        float4 a = *reinterpret_cast<const float4*>(xr);
        const float4* wb = reinterpret_cast<const float4*>(sB) + (4 * half) * NQ * 32 + col;   // + (8 kg + s) * NQ * 32 + 32 tq
        float4 wc[NQ];
#pragma unroll
        for (int tq = 0; tq < NQ; ++tq) wc[tq] = wb[32 * tq];
#pragma unroll
        for (int kg = 0; kg < K / 8; ++kg) {
            const float4 an = *reinterpret_cast<const float4*>(xr + 8 * (kg + 1 < K / 8 ? kg + 1 : kg));    // one group ahead
            const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int sstep = 0; sstep < 4; ++sstep) {
                const int nxt = kg * 8 + sstep + 1 < K - 4 ? (sstep < 3 ? kg * 8 + sstep + 1 : kg * 8 + 8) : kg * 8 + sstep;
                float4 wn[NQ];
#pragma unroll
                for (int tq = 0; tq < NQ; ++tq) wn[tq] = wb[nxt * NQ * 32 + 32 * tq];       // one step ahead
#pragma unroll
                for (int tq = 0; tq < NQ; ++tq) {
                    acc[4 * tq + 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[tq].x, av[sstep], acc[4 * tq + 0], 0, 0, 0);
                    acc[4 * tq + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[tq].y, av[sstep], acc[4 * tq + 1], 0, 0, 0);
                    acc[4 * tq + 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[tq].z, av[sstep], acc[4 * tq + 2], 0, 0, 0);
                    acc[4 * tq + 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[tq].w, av[sstep], acc[4 * tq + 3], 0, 0, 0);
                }
#pragma unroll
                for (int tq = 0; tq < NQ; ++tq) wc[tq] = wn[tq];
            }
            a = an;
        }
        // store: registers 4 q .. 4 q + 3 of tile t = features 32 t + 8 q + 4 half + (0..3) of this lane's point: float4
        if (valid) {
            float* yr = Y + prow * M + 4 * half;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int f0 = 32 * t + 8 * q + 4 * half;
                    float v[4] = {acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]};
                    float m4[4] = {mk[EPI == EPI_MASK || EPI == EPI_SEED ? t : 0][q].x, mk[EPI == EPI_MASK || EPI == EPI_SEED ? t : 0][q].y,
                                   mk[EPI == EPI_MASK || EPI == EPI_SEED ? t : 0][q].z, mk[EPI == EPI_MASK || EPI == EPI_SEED ? t : 0][q].w};
                    if (by_rec) {
                        const uint32_t wd = (t >> 1) == 0 ? rec.x : ((t >> 1) == 1 ? rec.y : ((t >> 1) == 2 ? rec.z : rec.w));
                        const uint32_t pat = (wd >> (16 * (t & 1))) & 0xffffu;
#pragma unroll
                        for (int e = 0; e < 4; ++e) m4[e] = ((pat >> (15 - (4 * q + e))) & 1u) ? 1.0f : 0.0f;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (EPI == EPI_BIAS_RELU) v[e] = fmaxf(v[e] + sV[f0 + e], 0.0f);
                        if (EPI == EPI_MASK) v[e] = m4[e] > 0.0f ? v[e] : 0.0f;
                        if (EPI == EPI_SEED) v[e] = m4[e] > 0.0f ? v[e] + scn * sV[f0 + e] : 0.0f;
                    }
                    *reinterpret_cast<float4*>(yr + 32 * t + 8 * q) = make_float4(v[0], v[1], v[2], v[3]);
                }
        }
    }
}
template <int K, int M, bool TRANS, int EPI>
void lin(const float* X, const float* W, float* Y, int64_t N, const float* bias_or_wv, const float* msrc, const float* sc, hipStream_t st,
         Rows rw = Rows{nullptr, nullptr}, const void* recs = nullptr, int rec_layer = 0) {
    const int64_t ntile = (N + 31) / 32;
    int groups = (int)((ntile + 3) / 4);
    if (groups > 256) groups = 256;               // one workgroup per CU: B is staged once per workgroup
    hipLaunchKernelGGL((k_t_lin<K, M, TRANS, EPI>), dim3((unsigned)groups), dim3(256), 0, st, X, W, Y, N, bias_or_wv, msrc, sc, rw,
                       (const uint4*)recs, rec_layer);
}

// k_t_lin16 (round 6): the split-fp16 sibling of k_t_lin<K, M, false, EPI_MASK | EPI_SEED> - the two data-gradient products of the
// heads' backward, d_hl1 = (hl1 > 0) (d_hl2 W_l2) and ahat_6 = m_6 (d_rr W_rgb1 + d_sig w_den).  On the fp32 MFMA they were bound by the
// matrix pipe (0.17 + 0.25 ms per 8192 x 64 step at 60 - 80 TFLOP/s); with hi + lo halves (X / sx and W, three K = 16 products per tile
// and step, as everywhere else in the backward) the pipe's share drops to a sixth and the sweeps run at what their 1.5 KB per row cost.
// sx: the batch-wide magnitude of X the producing sweep left (k_t_wcolsum's gmax).  W is split as it is staged: [k / 8][column] half8
// pairs, K M 4 bytes of LDS (128 KB for 128 x 256), one workgroup per CU.  A lane owns one row of the wave's 32-row tile and the k-groups
// of its half-wave: its 16 float4 of the NEXT tile are requested before the products of this one.  Same accumulator layout, same
// epilogues and stores as k_t_lin.
template <int K, int M, int EPI>
__global__ void __launch_bounds__(256, 1) k_t_lin16(const float* __restrict__ X, const float* __restrict__ sx_ptr, const float* __restrict__ W,
                                                    float* __restrict__ Y, int64_t N, const float* __restrict__ wv,
                                                    const float* __restrict__ msrc, const float* __restrict__ sc, Rows rw,
                                                    const uint4* __restrict__ recs, int rec_layer) {
    DSN_OWN_SIMD();
    static_assert(EPI == EPI_MASK || EPI == EPI_SEED, "the backward's two data-gradient products");
    constexpr int NT = M / 32, KG = K / 8, ST = K / 16;
    __shared__ __attribute__((aligned(16))) t_half8 sWh[KG][M];
    __shared__ __attribute__((aligned(16))) t_half8 sWl[KG][M];
    __shared__ float sV[M];
    const int tid = threadIdx.x;
    for (int i = tid; i < KG * M; i += 256) {
        const int kg = i / M, m = i % M;
        t_half8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float w = W[(8 * kg + j) * M + m];
            const _Float16 h = (_Float16)w;
            hi[j] = h;
            lo[j] = (_Float16)(w - (float)h);
        }
        sWh[kg][m] = hi;
        sWl[kg][m] = lo;
    }
    if (EPI == EPI_SEED)
        for (int i = tid; i < M; i += 256) sV[i] = wv[i];
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;
    const float sx = sx_ptr ? t_pow2_at_least(*sx_ptr) : 1.0f, inv = 1.0f / sx;
    const int64_t NL = rows_n(rw, N);
    const int64_t ntile = (NL + 31) / 32;
    const int64_t stride = (int64_t)gridDim.x * 4;
    float4 xn[2 * ST];
    auto request = [&](int64_t tile) {
        const int64_t lrow = tile * 32 + col;
        const float* xr = X + rows_at(rw, lrow < NL ? lrow : NL - 1) * K + 8 * half;
#pragma unroll
        for (int s = 0; s < ST; ++s) {
            xn[2 * s] = *reinterpret_cast<const float4*>(xr + 16 * s);
            xn[2 * s + 1] = *reinterpret_cast<const float4*>(xr + 16 * s + 4);
        }
    };
    int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    if (tile < ntile) request(tile);
    for (; tile < ntile; tile += stride) {
        const int64_t lrow = tile * 32 + col;
        const bool valid = lrow < NL;
        const int64_t crow = rows_at(rw, valid ? lrow : NL - 1);
        float4 xc[2 * ST];
#pragma unroll
        for (int i = 0; i < 2 * ST; ++i) xc[i] = xn[i];
        if (tile + stride < ntile) request(tile + stride);
        // what the epilogue needs from memory is requested now and used behind the products
        float4 mk[EPI == EPI_MASK ? NT : 1][4];
        float scn = 0.0f;
        const bool by_rec = EPI == EPI_SEED && recs != nullptr;      // kernel-uniform
        uint4 rec = make_uint4(0u, 0u, 0u, 0u);
        if (by_rec) rec = recs[((size_t)crow * 2 + half) * 7 + rec_layer];
        else if (EPI == EPI_MASK) {
            const float* mr = msrc + crow * M + 4 * half;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) mk[EPI == EPI_MASK ? t : 0][q] = *reinterpret_cast<const float4*>(mr + 32 * t + 8 * q);
        }
        if (EPI == EPI_SEED) scn = sc[crow];
        t_f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
#pragma unroll
        for (int s = 0; s < ST; ++s) {
            const float xv[8] = {xc[2 * s].x, xc[2 * s].y, xc[2 * s].z, xc[2 * s].w, xc[2 * s + 1].x, xc[2 * s + 1].y, xc[2 * s + 1].z, xc[2 * s + 1].w};
            t_half8 bh, bl;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float x = xv[j] * inv;
                const _Float16 h = (_Float16)x;
                bh[j] = h;
                bl[j] = (_Float16)(x - (float)h);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const t_half8 ah = sWh[2 * s + half][32 * t + col], al = sWl[2 * s + half][32 * t + col];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[t], 0, 0, 0);
            }
        }
        // store: registers 4 q .. 4 q + 3 of tile t = features 32 t + 8 q + 4 half + (0..3) of this lane's row: float4 (as k_t_lin)
        if (valid) {
            float* yr = Y + crow * M + 4 * half;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int f0 = 32 * t + 8 * q + 4 * half;
                    float v[4] = {acc[t][4 * q] * sx, acc[t][4 * q + 1] * sx, acc[t][4 * q + 2] * sx, acc[t][4 * q + 3] * sx};
                    const float4 m = mk[EPI == EPI_MASK ? t : 0][q];
                    float m4[4] = {m.x, m.y, m.z, m.w};
                    if (by_rec) {
                        const uint32_t wd = (t >> 1) == 0 ? rec.x : ((t >> 1) == 1 ? rec.y : ((t >> 1) == 2 ? rec.z : rec.w));
                        const uint32_t pat = (wd >> (16 * (t & 1))) & 0xffffu;
#pragma unroll
                        for (int e = 0; e < 4; ++e) m4[e] = ((pat >> (15 - (4 * q + e))) & 1u) ? 1.0f : 0.0f;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (EPI == EPI_MASK) v[e] = m4[e] > 0.0f ? v[e] : 0.0f;
                        if (EPI == EPI_SEED) v[e] = m4[e] > 0.0f ? v[e] + scn * sV[f0 + e] : 0.0f;
                    }
                    *reinterpret_cast<float4*>(yr + 32 * t + 8 * q) = make_float4(v[0], v[1], v[2], v[3]);
                }
        }
    }
}
template <int K, int M, int EPI>
void lin16(const float* X, const float* sx, const float* W, float* Y, int64_t N, const float* wv, const float* msrc, const float* sc,
           hipStream_t st, Rows rw, const void* recs = nullptr, int rec_layer = 0) {
    const int64_t ntile = (N + 31) / 32;
    int groups = (int)((ntile + 3) / 4);
    if (groups > 256) groups = 256;               // one workgroup per CU: W is split and staged once per workgroup
    hipLaunchKernelGGL((k_t_lin16<K, M, EPI>), dim3((unsigned)groups), dim3(256), 0, st, X, sx, W, Y, N, wv, msrc, sc, rw, (const uint4*)recs,
                       rec_layer);
}

struct TrainWs {
    uint8_t* transparent;
    int32_t* idx_c;
    float *x_c, *pe, *h[7], *ap[7], *tn[7], *an[7], *rr, *ess, *sig, *g, *t0, *tpe, *n_w, *xl, *hl1, *hl2, *pre, *wl, *col;
    void* masks;
    float *d_sig, *d_col, *d_ess, *d_pre, *d_hl2, *d_hl1, *d_xl, *d_rr, *u, *scratch_t, *small;
    float* wg_part;                // the workgroups' partial tiles of one split-fp16 weight-gradient product (k_t_wgrad_reduce)
    uint8_t* live;                 // [N] row flags (scratch of the list builds)
    int32_t *list1, *list2;        // [N] rows the forward evaluates / rows with non-zero cotangents (ascending sample indices)
    int32_t *bcnt;                 // [N / 256 + 1] per-block counts / offsets of a list build
    int32_t *rowcnt;               // [0] = entries of list1, [1] = entries of list2 (NOT in `small`: that is cleared per backward)
    size_t bytes;
};

TrainWs carve(void* base, int64_t N) {
    TrainWs w;
    char* p = (char*)base;
    auto take = [&](size_t bytes) { char* q = p; p += dsn_align256(bytes); return q; };
    const size_t n = (size_t)N;
    w.transparent = (uint8_t*)take(n);
    w.idx_c = (int32_t*)take(4 * n);
    w.x_c = (float*)take(12 * n);
    w.pe = (float*)take(4 * PE_LD * n);
    for (int l = 0; l < 7; ++l) w.h[l] = (float*)take(1024 * n);
    for (int l = 0; l < 7; ++l) w.ap[l] = (float*)take(1024 * n);
    for (int l = 0; l < 7; ++l) w.tn[l] = (float*)take(1024 * n);
    // (the adjoint pass's outputs: buffers of their own since round 5 - a layer's two weight-gradient products run as ONE launch, which
    //  reads the tangent array hdot_{l-1} and the adjoint array ahat_l side by side; rounds 1-4 wrote the adjoints over the tangents)
    for (int l = 0; l < 7; ++l) w.an[l] = (float*)take(1024 * n);
    w.masks = (void*)take(224 * n);
    w.rr = (float*)take(512 * n);
    w.ess = (float*)take(12 * n);
    w.sig = (float*)take(4 * n);
    w.g = (float*)take(12 * n);
    w.t0 = (float*)take(1024 * n);
    w.tpe = (float*)take(4 * PE_LD * n);
    w.n_w = (float*)take(12 * n);
    w.xl = (float*)take(36 * n);
    w.hl1 = (float*)take(512 * n);
    w.hl2 = (float*)take(512 * n);
    w.pre = (float*)take(4 * n);
    w.wl = (float*)take(4 * n);
    w.col = (float*)take(12 * n);
    w.d_sig = (float*)take(4 * n);
    w.d_col = (float*)take(12 * n);
    w.d_ess = (float*)take(12 * n);
    w.d_pre = (float*)take(4 * n);
    w.d_hl2 = (float*)take(512 * n);
    w.d_hl1 = (float*)take(512 * n);
    w.d_xl = (float*)take(36 * n);
    w.d_rr = (float*)take(512 * n);
    w.u = (float*)take(12 * n);
    w.scratch_t = (float*)take(4 * n);
    w.live = (uint8_t*)take(n);
    w.list1 = (int32_t*)take(4 * n);
    w.list2 = (int32_t*)take(4 * n);
    {   // (a product launches at most N / 64 workgroups: small batches keep a small workspace)
        size_t g = (n + 63) / 64;
        if (g > W16_PART_GROUPS) g = W16_PART_GROUPS;
        const size_t fl = g * (size_t)W16_PART(256) > (size_t)256 * W16_PART(256) ? (size_t)256 * W16_PART(256) : g * (size_t)W16_PART(256);
        const size_t fp = g * (size_t)W16_PART(64);
        w.wg_part = (float*)take(4 * (fl > fp ? fl : fp));
    }
    w.bcnt = (int32_t*)take(4 * (n / T_THREADS + 2));
    w.rowcnt = (int32_t*)take(256);           // (the 256 bytes in front of `small`: the host mirror reads the two counts there)
    w.small = (float*)take(4 * 1024);         // (last: the host mirror finds its counters at the end of the workspace)
    w.bytes = (size_t)(p - (char*)base);
    return w;
}

void colsum(const float* a, int C, int64_t N, float* out, hipStream_t st, Rows rw = Rows{nullptr, nullptr}) {
    const int rows = 256;
    hipLaunchKernelGGL(k_t_colsum, dim3((unsigned)((N + rows - 1) / rows)), dim3(T_THREADS), 0, st, a, C, N, rows, out, rw);
}

template <int OUT>
void wcolsum(const float* X, int C, const float* dY, int64_t N, float* dW, float* db, hipStream_t st, Rows rw = Rows{nullptr, nullptr},
             const float* W = nullptr, float* dX = nullptr, float* gmax = nullptr) {
    static const int rows_env = [] { const char* e = getenv("DSN_WCOLSUM_ROWS"); const int v = e ? atoi(e) : 0; return v >= 16 && v <= 4096 ? v : 0; }();
    const int rows = rows_env ? rows_env : 512;      // (round 6: 256 -> 512 rows per block, -0.07 ms per step - the blocks' atomics, not the rows, were the cost; 1024: slower)
    hipLaunchKernelGGL((k_t_wcolsum<OUT>), dim3((unsigned)((N + rows - 1) / rows)), dim3(T_THREADS), 0, st, X, C, dY, N, rows, dW, db, rw, W, dX,
                       (unsigned*)gmax);
}

}  // namespace

size_t dsn_train_workspace_size(int64_t N) { return carve(nullptr, N).bytes; }

DsnTrainCache dsn_train_cache(void* workspace, int64_t N) {
    const TrainWs w = carve(workspace, N);
    DsnTrainCache c = {w.transparent, w.idx_c, w.x_c, w.sig, w.ess, w.g, w.n_w, w.h[0], w.ap[0], w.rr, w.masks, w.hl1, w.hl2, w.pre,
                       w.live, w.list1, w.bcnt, w.rowcnt};
    return c;
}

// list <- ascending indices of the rows with flag != 0, *count <- their number (three small launches, deterministic)
// DSN_TRAIN_ALL_ROWS=1 (tests / A-B runs): every row is listed - the dense evaluation of round 2
static bool dsn_train_all_rows() { const char* e = getenv("DSN_TRAIN_ALL_ROWS"); return e && e[0] == '1'; }
void dsn_train_build_rows(uint8_t* flag, int64_t N, int32_t* bcnt, int32_t* list, int32_t* count, hipStream_t st) {
    if (dsn_train_all_rows()) (void)hipMemsetAsync(flag, 1, (size_t)N, st);
    const int nb = (int)((N + T_THREADS - 1) / T_THREADS);
    hipLaunchKernelGGL(k_t_rows_count, dim3((unsigned)nb), dim3(T_THREADS), 0, st, (const uint8_t*)flag, N, bcnt);
    hipLaunchKernelGGL(k_t_rows_scan, dim3(1), dim3(1024), 0, st, bcnt, nb, count);
    hipLaunchKernelGGL(k_t_rows_fill, dim3((unsigned)nb), dim3(T_THREADS), 0, st, (const uint8_t*)flag, N, (const int32_t*)bcnt, list);
}
// rows the training forward has to evaluate (see k_t_flag_forward) -> list, *count
void dsn_train_forward_rows(const uint8_t* transparent, const float* noise, int64_t N, uint8_t* flag, int32_t* bcnt, int32_t* list,
                            int32_t* count, hipStream_t st) {
    hipLaunchKernelGGL(k_t_flag_forward, grid_for(N), dim3(T_THREADS), 0, st, transparent, noise, N, flag);
    dsn_train_build_rows(flag, N, bcnt, list, count, st);
}

#define T_CHECK(x) do { if (!(x)) return #x; } while (0)

// returns nullptr on success, else a static description of the step that failed
// all 33 gradient tensors (and the small scratch) zeroed by ONE launch instead of 34 memsets (5 us each on the stream)
struct TrainZero { float* p[34]; int n[34]; };
__global__ void __launch_bounds__(T_THREADS) k_t_zero(TrainZero z) {
    float* __restrict__ p = z.p[blockIdx.y];
    const int n = z.n[blockIdx.y];
    for (int i = blockIdx.x * T_THREADS + threadIdx.x; i < n; i += gridDim.x * T_THREADS) p[i] = 0.0f;
}

const char* dsn_train_run(const DsnSceneView& s, const float* packed, const float* const* prm, const float* poses, int frame_idx,
                          int zero_code,
                          const float* ray_o, const float* ray_d, const float* z_vals, const float* noise, int R, int S,
                          const float* d_rgb, const float* d_disp, const float* d_acc, const float* d_depth,
                          const float* d_weights, float* const* grd, void* workspace, hipStream_t st, bool cached,
                          const float* ext_x_c, const float* ext_d_col, const float* ext_d_sig, const DsnTrainAux* aux) {
    // ext_* (all three or none): "module" mode, the backward of DualSpaceNeRF.forward (model/spacenet.py:210-266) on explicit
    // points - canonical points ext_x_c [N,3] instead of the warp of the rays' samples, and the per-sample cotangents of
    // (colour, density) ext_d_col [N,3] / ext_d_sig [N] instead of the adjoint of compositing.  The caller passes S = 1,
    // ray_o = world points, ray_d = view directions, z_vals = zeros (x_w = o + d * 0 exactly).
    const bool module = ext_x_c != nullptr;
    const int64_t N64 = (int64_t)R * S;
    if (N64 > (int64_t)1 << 30) return "batch too large for the 32-bit GEMM interface";
    TrainWs w = carve(workspace, N64);
    {
        TrainZero z;
        for (int i = 0; i < 33; ++i) { z.p[i] = grd[i]; z.n[i] = kParamCount[i]; }
        z.p[33] = w.small; z.n[33] = 1024;
        hipLaunchKernelGGL(k_t_zero, dim3(32, 34), dim3(T_THREADS), 0, st, z);
    }

    // ---- forward: warp, encoding, trunk, heads ------------------------------------------------------------
    // (skipped when dsn_render_rays_train has just left all of it in this workspace)
    // R1: the rows the forward evaluates (all but transparent samples with noise <= 0: alpha = 0 exactly); R2 (below): the rows
    // with non-zero cotangents - the only ones that reach a gradient.  Arrays stay indexed by sample.
    Rows R1 = {nullptr, nullptr};
    if (module) {
        if (hipMemcpyAsync(w.x_c, ext_x_c, sizeof(float) * 3 * (size_t)N64, hipMemcpyDeviceToDevice, st) != hipSuccess) return "x_c copy";
    } else {
        if (!cached) {
            dsn_launch_warp(s, nullptr, ray_o, ray_d, z_vals, N64, S, nullptr, nullptr, nullptr, w.transparent, w.x_c, nullptr, nullptr,
                            nullptr, false, st);
            dsn_train_forward_rows(w.transparent, noise, N64, w.live, w.bcnt, w.list1, w.rowcnt, st);
        }
        R1 = Rows{w.list1, w.rowcnt};
    }
    // trunk + heads forward and the sigma reverse pass in ONE fused split-fp16 launch (k_field16<train>): besides sigma,
    // essence and g = d sigma/dx it leaves every layer's activations h_l, the masked sigma-adjoints a_l and the rgb hidden
    // layer in the row-major arrays the weight-gradient products below read
    const dim3 wave_grid((unsigned)((N64 + 3) / 4));
    if (!cached) dsn_launch_field16_train(packed, s.frame, w.x_c, N64, w.sig, w.ess, w.g, w.h[0], w.ap[0], w.rr, w.masks, st, nullptr,
                                          R1.list, R1.cnt);

    // ---- normals, lighting, colour ---------------------------------------------------------------------------
    if (!cached) dsn_launch_normal(s, w.x_c, w.g, N64, R1.list, R1.cnt, w.idx_c, w.n_w, false, st);
    hipLaunchKernelGGL(k_t_light_in, grid_for(N64), dim3(T_THREADS), 0, st, w.n_w, ray_o, ray_d, z_vals, s.frame, N64, S, w.xl, R1);
    if (!cached) {      // (the training forward's k_light16 has left hl1, hl2 and pre in this workspace otherwise)
        hipLaunchKernelGGL(k_t_light_first, dim3((unsigned)((N64 + 255) / 256)), dim3(T_THREADS), 0, st, w.xl, prm[P_L0_W], prm[P_L0_B],
                           N64, 256, w.hl1, R1);
        lin<128, 128, true, EPI_BIAS_RELU>(w.hl1, prm[P_L2_W], w.hl2, N64, prm[P_L2_B], nullptr, nullptr, st, R1);   // hl2 = relu(hl1 W2^T + b2)
        hipLaunchKernelGGL(k_t_rowdot, wave_grid, dim3(T_THREADS), 0, st, w.hl2, 128, prm[P_L4_W], prm[P_L4_B], 1, N64, w.pre, R1);
    }
    hipLaunchKernelGGL(k_t_colour, grid_for(N64), dim3(T_THREADS), 0, st, w.pre, w.ess, N64, w.wl, w.col, R1);

    // ---- adjoint of compositing and of the colour product ------------------------------------------------------
    if (!module) {
        // one wave per ray for S <= 128 (round 6); DSN_TRAIN_COMPOSITE_ADJOINT=thread keeps the one-thread-per-ray form (A/B, cross-check)
        static const bool per_thread = [] { const char* e = getenv("DSN_TRAIN_COMPOSITE_ADJOINT"); return e && e[0] == 't'; }();
        const dim3 wg((unsigned)((R + T_THREADS / 64 - 1) / (T_THREADS / 64)));
        if (!per_thread && S <= 64)
            hipLaunchKernelGGL(k_t_composite_adjoint_w<1>, wg, dim3(T_THREADS), 0, st, w.col, w.sig, w.transparent, z_vals, ray_d, noise, R, S,
                               d_rgb, d_disp, d_acc, d_depth, d_weights, w.d_col, w.d_sig, w.live);
        else if (!per_thread && S <= 128)
            hipLaunchKernelGGL(k_t_composite_adjoint_w<2>, wg, dim3(T_THREADS), 0, st, w.col, w.sig, w.transparent, z_vals, ray_d, noise, R, S,
                               d_rgb, d_disp, d_acc, d_depth, d_weights, w.d_col, w.d_sig, w.live);
        else
            hipLaunchKernelGGL(k_t_composite_adjoint, grid_for(R), dim3(T_THREADS), 0, st, w.col, w.sig, w.transparent, z_vals, ray_d,
                               noise, R, S, d_rgb, d_disp, d_acc, d_depth, d_weights, w.scratch_t, w.d_col, w.d_sig, w.live);
    }
    const float* const d_col = module ? ext_d_col : w.d_col;
    const float* const d_sig = module ? ext_d_sig : w.d_sig;
    if (module) hipLaunchKernelGGL(k_t_flag_cotangent, grid_for(N64), dim3(T_THREADS), 0, st, d_col, d_sig, N64, w.live);
    dsn_train_build_rows(w.live, N64, w.bcnt, w.list2, w.rowcnt + 1, st);
    const Rows R2 = {w.list2, w.rowcnt + 1};
    // (the encoding of the backward's rows is written by k_t_pe_tangent below, beside its tangent; DSN_TRAIN_UNFUSED_HEADS=1: by k_t_pe here)
    static const bool two_pass_heads = [] { const char* e = getenv("DSN_TRAIN_UNFUSED_HEADS"); return e && e[0] == '1'; }();
    static const bool heads_fp32 = [] { const char* e = getenv("DSN_TRAIN_HEADS"); return e && e[0] == 'f'; }();
    const bool heads16 = !heads_fp32 && !two_pass_heads;      // (the scales ride in the fused sweeps)
    float* const g_dhl2 = w.small + 304;      // batch-wide max |d_hl2|, |d_rr| (float bits; zeroed with w.small)
    float* const g_drr = w.small + 305;
    if (two_pass_heads) hipLaunchKernelGGL(k_t_pe, grid_for(N64 * PE_LD), dim3(T_THREADS), 0, st, w.x_c, N64, w.pe, R2);
    hipLaunchKernelGGL(k_t_colour_adjoint, grid_for(N64), dim3(T_THREADS), 0, st, d_col, w.ess, w.wl, w.pre, N64, w.d_ess,
                       w.d_pre, R2);
    // Two chains that do not depend on each other hang off this point (round 6): T - the lighting MLP's backward, u = dL/dg, the PE
    // tangent and k_tangent16 - and A - the colour head's backward, the adjoint seed and k_adjoint16.  Most of their kernels are small
    // (0.1 - 0.25 ms, a few hundred MB each, 2 - 3 TB/s alone): with a second stream of the CALLER's (dsn_render_rays_grad_ex: aux
    // stream + two events, fork here, join in front of the weight-gradient products) chain A runs beside chain T.  Same kernels, same
    // arguments, same values; the only shared scratch - the partial tiles of the two 128-row products - is given a place of its own
    // (three adjoint layers' arrays, which k_adjoint16 fills later on the same stream).  Without aux: one stream, as before.
    static const bool pairs_ = [] { const char* e = getenv("DSN_WGRAD_PAIRS"); return !(e && e[0] == '0'); }();
    const bool two = aux && aux->stream && aux->fork && aux->join && pairs_ && heads16 && N64 >= 4096;
    hipStream_t sa = st;
    if (two) {
        sa = aux->stream;
        if (hipEventRecord(aux->fork, st) != hipSuccess || hipStreamWaitEvent(sa, aux->fork, 0) != hipSuccess) return "fork of the auxiliary stream";
    }
    float* const part_a = two ? w.an[3] : w.wg_part;      // (3 KB per row: an[3 .. 5], contiguous; a 128 x 256 tile per workgroup needs 2.1 KB per row at most)

    // ---- lighting MLP backward -----------------------------------------------------------------------------------
    // (the heads' weight gradients and the data gradients behind their relus in ONE sweep over hl2 / rr; DSN_TRAIN_UNFUSED_HEADS=1: two)
    if (two_pass_heads) {
        wcolsum<1>(w.hl2, 128, w.d_pre, N64, grd[P_L4_W], grd[P_L4_B], st, R2);
        hipLaunchKernelGGL(k_t_seed, grid_for((N64 * 128) / 4), dim3(T_THREADS), 0, st, w.hl2, prm[P_L4_W], w.d_pre, nullptr, 128, N64 * 128,
                           w.d_hl2, R2);
    } else
        wcolsum<1>(w.hl2, 128, w.d_pre, N64, grd[P_L4_W], grd[P_L4_B], st, R2, prm[P_L4_W], w.d_hl2, g_dhl2);
    // (round 6: the 128-wide products of the lighting MLP and the colour head on the split-fp16 kernel k_t_wgrad16q - scales: the batch-wide
    //  magnitudes the sweeps above leave; DSN_TRAIN_HEADS=fp32 keeps the exact-fp32 MFMA kernel, A/B and cross-check)
    if (heads16) wgrad_mfma16q<128>(N64, w.hl1, nullptr, w.d_hl2, g_dhl2, grd[P_L2_W], 128, st, grd[P_L2_B], R2, w.wg_part);
    else T_CHECK(wgrad_mfma(N64, 128, 128, 128, w.hl1, 128, w.d_hl2, 128, grd[P_L2_W], 128, st, grd[P_L2_B], R2));
    // (round 6: the two data-gradient products of the heads on the split-fp16 kernel k_t_lin16; DSN_TRAIN_LIN=fp32 keeps k_t_lin, A/B and cross-check)
    static const bool lin_fp32 = [] { const char* e = getenv("DSN_TRAIN_LIN"); return e && e[0] == 'f'; }();
    const bool lin16_ = heads16 && !lin_fp32;      // (the operand scales come from the fused sweeps)
    if (lin16_) lin16<128, 128, EPI_MASK>(w.d_hl2, g_dhl2, prm[P_L2_W], w.d_hl1, N64, nullptr, w.hl1, nullptr, st, R2);   // d_hl1 = (hl1 > 0) (d_hl2 W2)
    else lin<128, 128, false, EPI_MASK>(w.d_hl2, prm[P_L2_W], w.d_hl1, N64, nullptr, w.hl1, nullptr, st, R2);
    // (the first lighting layer keeps its two passes over d_hl1: a VALU kernel that also accumulates dW0 takes 0.39 ms against
    //  0.107 for the exact-fp32 MFMA product + 0.129 for the data gradient - tried in round 5)
    if (!wgrad_mfma(N64, 32, 9, 128, w.xl, 9, w.d_hl1, 128, grd[P_L0_W], 9, st, grd[P_L0_B], R2)) {
        if (two) (void)hipStreamSynchronize(sa);      // (a failed launch leaves no chain running on the caller's second stream behind an error)
        return "wgrad_mfma(lights_encoding.0)";
    }
    hipLaunchKernelGGL(k_t_light_first_bwd, dim3((unsigned)((N64 + 63) / 64)), dim3(T_THREADS), 0, st, w.d_hl1, prm[P_L0_W], N64,
                       w.d_xl, R2);

    // ---- u = dL/dg through the normal map, then the tangent pass (second-order term) ---------------------------
    hipLaunchKernelGGL(k_t_normal_adjoint, grid_for(N64), dim3(T_THREADS), 0, st, s.face_world, s.face_canon, w.x_c, w.g, w.idx_c,
                       w.d_xl, N64, w.u, R2);
    float* const g_tpe = w.small + 302;        // batch-wide max |tpe| (float bits; zeroed with w.small)
    {
        const int64_t nb = (N64 * PE_LD + T_THREADS - 1) / T_THREADS;
        hipLaunchKernelGGL(k_t_pe_tangent, dim3((unsigned)(nb < 4096 ? nb : 4096)), dim3(T_THREADS), 0, st, w.x_c, w.u, N64, w.tpe,
                           (unsigned*)g_tpe, R2, two_pass_heads ? nullptr : w.pe);
    }
    // all seven tangent layers in one fused split-fp16 launch (k_tangent16, relu patterns from the training forward's records),
    // then the weight-gradient products  dW_l += a_l^T hdot_{l-1}
    float* const g_tan = w.small + 300;        // batch-wide magnitudes of the tangent / adjoint arrays (zeroed with w.small)
    float* const g_adj = w.small + 301;
    int32_t* const range_cnt = (int32_t*)(w.small + 303);      // samples whose tangent / adjoint left the fp16 range (zeroed with w.small)
    // (round 6: the last layer's tangent is only ever summed over the rows - d (w_d . hdot_6) / d w_d - so k_tangent16 sums it itself
    //  and does not store it; DSN_TRAIN_COLSUM6=kernel keeps the stored layer + k_t_colsum, A/B and cross-check)
    static const bool colsum6_apart = [] { const char* e = getenv("DSN_TRAIN_COLSUM6"); return e && e[0] == 'k'; }();
    dsn_launch_tangent16(packed, w.x_c, w.u, N64, w.masks, w.tn[0], g_tan, st, range_cnt, R2.list, R2.cnt, colsum6_apart ? nullptr : grd[P_DEN_W]);
    // DSN_WGRAD_PAIRS=0 (A/B, cross-check): rounds 1-4's form - a launch per product, the adjoints written over the tangent arrays
    static const bool pairs = [] { const char* e = getenv("DSN_WGRAD_PAIRS"); return !(e && e[0] == '0'); }();
    if (!pairs) {
        wgrad_mfma16p(N64, w.tpe, g_tpe, w.ap[0], nullptr, grd[P_S1_0W] + W0_PE_COL, 87, PE_K, st, nullptr, R2, w.wg_part);
        for (int l = 1; l < 7; ++l)
            wgrad_mfma16(N64, w.tn[l - 1], g_tan, w.ap[l], nullptr, grd[kTrunkW[l]], kTrunkLd[l], st, nullptr, R2, w.wg_part);
        wgrad_mfma16p(N64, w.tpe, g_tpe, w.ap[4], nullptr, grd[P_S2_0W] + W4_PE_COL, 319, PE_K, st, nullptr, R2, w.wg_part);
    }
    // (default: the products  dW_l += a_l^T hdot_{l-1}  wait for the adjoint pass and run in ONE launch per layer with  ahat_l^T h_{l-1})
    if (colsum6_apart) colsum(w.tn[6], 256, N64, grd[P_DEN_W], st, R2);   // d (w_d . hdot_6) / d w_d
    float* cur = w.t0;

    // ---- adjoint pass of dL/dsigma * sigma + dL/dessence . essence ---------------------------------------------
    if (two_pass_heads) {
        wcolsum<3>(w.rr, 128, w.d_ess, N64, grd[P_RGB3_W], grd[P_RGB3_B], st, R2);
        hipLaunchKernelGGL(k_t_rgb_hidden_adjoint, grid_for((N64 * 128) / 4), dim3(T_THREADS), 0, st, w.d_ess, prm[P_RGB3_W], w.rr, N64 * 128,
                           w.d_rr, R2);
    } else
        wcolsum<3>(w.rr, 128, w.d_ess, N64, grd[P_RGB3_W], grd[P_RGB3_B], sa, R2, prm[P_RGB3_W], w.d_rr, g_drr);
    if (heads16) wgrad_mfma16q<256>(N64, w.h[6], nullptr, w.d_rr, g_drr, grd[P_RGB1_W], 256, sa, grd[P_RGB1_B], R2, part_a);
    else T_CHECK(wgrad_mfma(N64, 256, 256, 128, w.h[6], 256, w.d_rr, 128, grd[P_RGB1_W], 256, st, grd[P_RGB1_B], R2));
    wcolsum<1>(w.h[6], 256, d_sig, N64, grd[P_DEN_W], grd[P_DEN_B], sa, R2);
    // cur = ahat_6 = (h6 > 0) (d_rr W_rgb1 + d_sig w_den): the colour head's data gradient with the density head's seed fused in
    // (round 6: the relu pattern of layer 6 from the forward's records - 32 bytes per row instead of the 1 KB row of h_6;
    //  DSN_TRAIN_SEED_MASK=h keeps the `h_6 > 0` form, A/B and cross-check)
    static const bool seed_by_h = [] { const char* e = getenv("DSN_TRAIN_SEED_MASK"); return e && e[0] == 'h'; }();
    if (lin16_ && !seed_by_h) lin16<128, 256, EPI_SEED>(w.d_rr, g_drr, prm[P_RGB1_W], cur, N64, prm[P_DEN_W], nullptr, d_sig, sa, R2, w.masks, 6);   // (pattern from the records only)
    else lin<128, 256, false, EPI_SEED>(w.d_rr, prm[P_RGB1_W], cur, N64, prm[P_DEN_W], w.h[6], d_sig, sa, R2, seed_by_h ? nullptr : w.masks, 6);
    // cur = ahat_6.  The layers below it in one fused split-fp16 launch (k_adjoint16 -> ahat_5 ... ahat_0 in the buffers the
    // tangent products are done with), then  dW_l += ahat_l^T h_{l-1}  and the bias gradients (column sums)
    float* const* an = pairs ? w.an : w.tn;
    dsn_launch_adjoint16(packed, N64, w.masks, cur, an[0], g_adj, sa, range_cnt, R2.list, R2.cnt);
    if (two && (hipEventRecord(aux->join, sa) != hipSuccess || hipStreamWaitEvent(st, aux->join, 0) != hipSuccess)) return "join of the auxiliary stream";
    // The first layer's product FIRST (the order of the products is free: every one has outputs of its own): what hangs on its bias
    // column sums - the stage1.0 bias copy and two single-workgroup kernels (constant input columns, embedding row, pose code ->
    // pose_mlp: 0.05 ms of an idle chip at the end of the step until the last session of round 6) - then runs on the auxiliary stream
    // beside the other seven products and is joined behind them.
    // (the bias gradient of stage1.0 = column sums of ahat_0 rides along into w.small[0..255])
    if (pairs)
        wgrad_mfma16p(N64, w.tpe, g_tpe, w.ap[0], nullptr, grd[kTrunkW[0]] + W0_PE_COL, 87, PE_K, st, w.small, R2, w.wg_part,
                      WgradPair{w.pe, nullptr, an[0], g_adj});
    else
        wgrad_mfma16p(N64, w.pe, nullptr, an[0], g_adj, grd[kTrunkW[0]] + W0_PE_COL, 87, PE_K, st, w.small, R2, w.wg_part);
    bool tail_forked = false;
    {
        hipStream_t sx = st;
        if (two) {
            if (hipEventRecord(aux->fork, st) != hipSuccess || hipStreamWaitEvent(sa, aux->fork, 0) != hipSuccess) return "fork of the auxiliary stream (first-layer tail)";
            sx = sa;
            tail_forked = true;
        }
        // stage1.0 bias, constant input columns, embedding row, pose code -> pose_mlp
        const bool copied = hipMemcpyAsync(grd[P_S1_0B], w.small, 256 * sizeof(float), hipMemcpyDeviceToDevice, sx) == hipSuccess;
        if (copied) {
            hipLaunchKernelGGL(k_t_first_layer_consts, dim3(1), dim3(256), 0, sx, w.small, prm[P_S1_0W], s.frame, frame_idx, zero_code,
                               grd[P_S1_0W], grd[P_EMB], w.small + 256);
            hipLaunchKernelGGL(k_t_pose_mlp_adjoint, dim3(1), dim3(256), 0, sx, prm[P_PM0_W], prm[P_PM0_B], prm[P_PM2_W], prm[P_PM2_B],
                               prm[P_PM4_W], poses, w.small + 256, grd[P_PM0_W], grd[P_PM0_B], grd[P_PM2_W], grd[P_PM2_B], grd[P_PM4_W],
                               grd[P_PM4_B]);
        }
        if (!copied) {      // (an error between fork and join must not leave the auxiliary stream running beside the caller)
            if (tail_forked) { (void)hipEventRecord(aux->join, sa); (void)hipStreamWaitEvent(st, aux->join, 0); }
            return "bias copy";
        }
    }
    for (int l = 6; l >= 1; --l) {
        const float* A = l == 6 ? cur : an[l];
        if (pairs) {
            // tangent pair (a_l, hdot_{l-1}) first, adjoint pair (ahat_l, h_{l-1}) second: one launch, one reduction
            wgrad_mfma16(N64, w.tn[l - 1], g_tan, w.ap[l], nullptr, grd[kTrunkW[l]], kTrunkLd[l], st, grd[kTrunkB[l]], R2, w.wg_part,
                         WgradPair{w.h[l - 1], nullptr, A, g_adj});
            if (l == 4)
                wgrad_mfma16p(N64, w.tpe, g_tpe, w.ap[4], nullptr, grd[kTrunkW[4]] + W4_PE_COL, 319, PE_K, st, nullptr, R2, w.wg_part,
                              WgradPair{w.pe, nullptr, A, g_adj});
            continue;
        }
        wgrad_mfma16(N64, w.h[l - 1], nullptr, A, g_adj, grd[kTrunkW[l]], kTrunkLd[l], st, grd[kTrunkB[l]], R2, w.wg_part);
        if (l == 4) wgrad_mfma16p(N64, w.pe, nullptr, A, g_adj, grd[kTrunkW[4]] + W4_PE_COL, 319, PE_K, st, nullptr, R2, w.wg_part);
    }
    if (tail_forked && (hipEventRecord(aux->join, sa) != hipSuccess || hipStreamWaitEvent(st, aux->join, 0) != hipSuccess))
        return "join of the auxiliary stream (first-layer tail)";
    return nullptr;
}
