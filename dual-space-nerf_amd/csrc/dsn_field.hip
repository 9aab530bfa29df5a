// dsn_field.hip - the networks of the hot path on the gfx950 matrix cores.
//
//   k_field : positional encoding -> stage1 -> stage2 -> density / colour heads AND the analytic
//             reverse pass d sigma / d x_c (model/spacenet.py:93-148 + :301-311), one launch.
//   k_light : LightingMLP (model/spacenet.py:174-188, :254-265).
//
// Formulation (CDNA4-first, not a GEMM library call): everything is computed TRANSPOSED,
//   H_out^T [features x points] = W [features x k] * H_in^T [k x points],
// with v_mfma_f32_32x32x2_f32 (exact fp32, k-ordered fma chain): the weights are the A operand,
// the activations the B operand.  One wavefront owns 32 sample points for the whole network.  The
// accumulator layout of the 32x32 MFMA (lane l, reg r <-> row (r&3)+8(r>>2)+4(l>>5), col l&31) is,
// register for register, a legal B-operand layout (lane l supplies B[k = l>>5][col = l&31]) as long
// as the k index is permuted consistently - and a contraction does not care about the order of k.
// So the weights are pre-permuted once (dsn_pack_params) and the activations of all 8 trunk layers,
// the heads and the 8 reverse-mode layers never leave the register file: no LDS round trip, no
// cross-lane shuffles, no HBM traffic between layers.  Per 32 points: 13 824 MFMAs (884 736 matrix
// cycles) against 12 B in / 28 B out per point; weights (3.4 MB) stream from L2 as the A operand,
// one coalesced 16-byte load per lane per 4 MFMAs.
#include "dsn_common.h"
#include "dsn_kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define DSN_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// ---------------------------------------------------------------------------------------------
// parameter packing
// ---------------------------------------------------------------------------------------------
struct DsnParamPtrs { const float* p[DSN_NUM_PARAMS_INTERNAL]; };

enum { P_EMB = 0, P_S1_0W, P_S1_0B, P_S1_2W, P_S1_2B, P_S1_4W, P_S1_4B, P_S1_6W, P_S1_6B, P_S2_0W, P_S2_0B, P_S2_2W,
       P_S2_2B, P_S2_4W, P_S2_4B, P_DEN_W, P_DEN_B, P_RGB1_W, P_RGB1_B, P_RGB3_W, P_RGB3_B, P_L0_W, P_L0_B, P_L2_W,
       P_L2_B, P_L4_W, P_L4_B, P_PM0_W, P_PM0_B, P_PM2_W, P_PM2_B, P_PM4_W, P_PM4_B };

enum { IMG_FWD = 0, IMG_BWD = 1, IMG_LT0 = 2, IMG_COPY = 3 };
struct DsnImage { int dst, count, src, ld, col0, MB, KB, kind, pe_from; };
// pe_from: forward images: first kb that is a positional-encoding block (KB = none);
//          transposed images: first m block that is a positional-encoding block (MB = none);
//          pe columns start at col0 + (IMG_FWD ? 256*(pe_from==8) : ...) - given explicitly below via pecol.
struct DsnImageX { DsnImage im; int pecol; };

#define DSN_IMAGE_TABLE \
    {{OFF_L0,   8 * 2 * DSN_BLK,  P_S1_0W,  87,  0,   8,  2,  IMG_FWD, 0},  8}, \
    {{OFF_L1,   8 * 8 * DSN_BLK,  P_S1_2W,  256, 0,   8,  8,  IMG_FWD, 8},  0}, \
    {{OFF_L2,   8 * 8 * DSN_BLK,  P_S1_4W,  256, 0,   8,  8,  IMG_FWD, 8},  0}, \
    {{OFF_L3,   8 * 8 * DSN_BLK,  P_S1_6W,  256, 0,   8,  8,  IMG_FWD, 8},  0}, \
    {{OFF_L4,   8 * 10 * DSN_BLK, P_S2_0W,  319, 0,   8,  10, IMG_FWD, 8},  256}, \
    {{OFF_L5,   8 * 8 * DSN_BLK,  P_S2_2W,  256, 0,   8,  8,  IMG_FWD, 8},  0}, \
    {{OFF_L6,   8 * 8 * DSN_BLK,  P_S2_4W,  256, 0,   8,  8,  IMG_FWD, 8},  0}, \
    {{OFF_RGB1, 4 * 8 * DSN_BLK,  P_RGB1_W, 256, 0,   4,  8,  IMG_FWD, 8},  0}, \
    {{OFF_L6T,  8 * 8 * DSN_BLK,  P_S2_4W,  256, 0,   8,  8,  IMG_BWD, 8},  0}, \
    {{OFF_L5T,  8 * 8 * DSN_BLK,  P_S2_2W,  256, 0,   8,  8,  IMG_BWD, 8},  0}, \
    {{OFF_L4T,  10 * 8 * DSN_BLK, P_S2_0W,  319, 0,   10, 8,  IMG_BWD, 8},  256}, \
    {{OFF_L3T,  8 * 8 * DSN_BLK,  P_S1_6W,  256, 0,   8,  8,  IMG_BWD, 8},  0}, \
    {{OFF_L2T,  8 * 8 * DSN_BLK,  P_S1_4W,  256, 0,   8,  8,  IMG_BWD, 8},  0}, \
    {{OFF_L1T,  8 * 8 * DSN_BLK,  P_S1_2W,  256, 0,   8,  8,  IMG_BWD, 8},  0}, \
    {{OFF_L0T,  2 * 8 * DSN_BLK,  P_S1_0W,  87,  0,   2,  8,  IMG_BWD, 0},  8}, \
    {{OFF_LT0,  4 * 1 * DSN_BLK,  P_L0_W,   9,   0,   4,  1,  IMG_LT0, 1},  0}, \
    {{OFF_LT1,  4 * 4 * DSN_BLK,  P_L2_W,   128, 0,   4,  4,  IMG_FWD, 4},  0}, \
    {{OFF_B1 + 0 * 256, 256, P_S1_2B, 0, 0, 0, 0, IMG_COPY, 0}, 0}, \
    {{OFF_B1 + 1 * 256, 256, P_S1_4B, 0, 0, 0, 0, IMG_COPY, 0}, 0}, \
    {{OFF_B1 + 2 * 256, 256, P_S1_6B, 0, 0, 0, 0, IMG_COPY, 0}, 0}, \
    {{OFF_B1 + 3 * 256, 256, P_S2_0B, 0, 0, 0, 0, IMG_COPY, 0}, 0}, \
    {{OFF_B1 + 4 * 256, 256, P_S2_2B, 0, 0, 0, 0, IMG_COPY, 0}, 0}, \
    {{OFF_B1 + 5 * 256, 256, P_S2_4B, 0, 0, 0, 0, IMG_COPY, 0}, 0}, \
    {{OFF_BRGB1, 128, P_RGB1_B, 0, 0, 0, 0, IMG_COPY, 0}, 0}, \
    {{OFF_WDEN, 256, P_DEN_W, 0, 0, 0, 0, IMG_COPY, 0}, 0}, \
    {{OFF_WRGB3, 384, P_RGB3_W, 0, 0, 0, 0, IMG_COPY, 0}, 0}, \
    {{OFF_BLT0, 128, P_L0_B, 0, 0, 0, 0, IMG_COPY, 0}, 0}, \
    {{OFF_BLT1, 128, P_L2_B, 0, 0, 0, 0, IMG_COPY, 0}, 0}, \
    {{OFF_WLT2, 128, P_L4_W, 0, 0, 0, 0, IMG_COPY, 0}, 0}, \
    {{OFF_SCAL + 0, 1, P_DEN_B, 0, 0, 0, 0, IMG_COPY, 0}, 0}, \
    {{OFF_SCAL + 1, 3, P_RGB3_B, 0, 0, 0, 0, IMG_COPY, 0}, 0}, \
    {{OFF_SCAL + 4, 1, P_L4_B, 0, 0, 0, 0, IMG_COPY, 0}, 0}, \
    {{OFF_RAW_W0, 256 * 87, P_S1_0W, 0, 0, 0, 0, IMG_COPY, 0}, 0}, \
    {{OFF_RAW_B0, 256, P_S1_0B, 0, 0, 0, 0, IMG_COPY, 0}, 0}, \
    {{OFF_RAW_EMB, 500 * 8, P_EMB, 0, 0, 0, 0, IMG_COPY, 0}, 0}, \
    {{OFF_RAW_PM0W, 64 * 92, P_PM0_W, 0, 0, 0, 0, IMG_COPY, 0}, 0}, \
    {{OFF_RAW_PM0B, 64, P_PM0_B, 0, 0, 0, 0, IMG_COPY, 0}, 0}, \
    {{OFF_RAW_PM2W, 64 * 64, P_PM2_W, 0, 0, 0, 0, IMG_COPY, 0}, 0}, \
    {{OFF_RAW_PM2B, 64, P_PM2_B, 0, 0, 0, 0, IMG_COPY, 0}, 0}, \
    {{OFF_RAW_PM4W, 16 * 64, P_PM4_W, 0, 0, 0, 0, IMG_COPY, 0}, 0}, \
    {{OFF_RAW_PM4B, 16, P_PM4_B, 0, 0, 0, 0, IMG_COPY, 0}, 0}, \

__constant__ DsnImageX g_images[] = {DSN_IMAGE_TABLE};
static const DsnImageX h_images[] = {DSN_IMAGE_TABLE};
#define DSN_NUM_IMAGES ((int)(sizeof(g_images) / sizeof(g_images[0])))

// weight the A operand of block (mb, kb) holds for k-slot r (0..15) of `lane` - shared by the fp32 image
// (r = 4*r4 + j) and the split-fp16 image (r = 8*t + j): see the header comment for the lane mapping
__host__ __device__ inline float dsn_weight_at(const DsnImageX& x, const float* __restrict__ src, int mb, int kb, int r, int lane) {
    const DsnImage& im = x.im;
    const int half = lane >> 5, row = lane & 31;
    if (im.kind == IMG_LT0) {
        int k = 2 * r + half;
        return (r < 5 && k < 9) ? src[(32 * mb + row) * im.ld + k] : 0.0f;
    }
    if (im.kind == IMG_FWD) {
        int out = 32 * mb + row;
        if (kb >= im.pe_from) {
            int pi = dsn_pe_index(16 * (kb - im.pe_from) + r, half);
            return pi < 0 ? 0.0f : src[out * im.ld + x.pecol + pi];
        }
        return src[out * im.ld + im.col0 + 32 * kb + dsn_crow(r, half)];
    }
    // IMG_BWD: rows = input features, k = output features
    int o = 32 * kb + dsn_crow(r, half);
    if (mb >= im.pe_from) {
        int t = 16 * (mb - im.pe_from) + (row & 3) + 4 * (row >> 3), hf = (row >> 2) & 1;
        int pi = dsn_pe_index(t, hf);
        return pi < 0 ? 0.0f : src[o * im.ld + x.pecol + pi];
    }
    return src[o * im.ld + im.col0 + 32 * mb + row];
}

// value of packed element e of the fp32 image x
__host__ __device__ inline float dsn_pack_value(const DsnImageX& x, const float* __restrict__ src, int e) {
    const DsnImage& im = x.im;
    if (im.kind == IMG_COPY) return src[e];
    int blk = e / DSN_BLK, w = e % DSN_BLK;
    int r4 = w / 256, lane = (w % 256) / 4, j = w % 4;
    return dsn_weight_at(x, src, blk / im.KB, blk % im.KB, 4 * r4 + j, lane);
}

// halfword hw (0..2047) of the split-fp16 image of stream block (image x, local block blk)
__host__ __device__ inline _Float16 dsn_pack_value16(const DsnImageX& x, const float* __restrict__ src, int blk, int hw) {
    const DsnImage& im = x.im;
    const int t = hw >> 10, part = (hw >> 9) & 1, lane = (hw >> 3) & 63, j = hw & 7;
    const float v = dsn_weight_at(x, src, blk / im.KB, blk % im.KB, 8 * t + j, lane);
    const _Float16 hi = (_Float16)v;
    if (x.im.dst < OFF_L6T) {
        // forward trunk + rgb head images (k_field16 forward pass): the whole image carries 2^6 - hi and lo share one
        // accumulator, and 2^6 keeps the residual of any weight >= 2^-8 a normal fp16 (weights up to 1023 fit)
        if (!part) return (_Float16)((float)hi * 64.0f);
        return (_Float16)((v - (float)hi) * 64.0f);
    }
    if (!part) return hi;
    return (_Float16)((v - (float)hi) * DSN_LO_SCALE);
}

__global__ void k_pack_params(DsnParamPtrs pp, float* __restrict__ packed) {
    const DsnImageX x = g_images[blockIdx.y];
    const float* src = pp.p[x.im.src];
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < x.im.count; e += gridDim.x * blockDim.x)
        packed[x.im.dst + e] = dsn_pack_value(x, src, e);
}

// split-fp16 stream: image i (first 15 table entries, stream order) owns blocks [dst/DSN_BLK, (dst+count)/DSN_BLK)
__global__ void k_pack_params16(DsnParamPtrs pp, _Float16* __restrict__ dst16) {
    const DsnImageX x = g_images[blockIdx.y];
    const float* src = pp.p[x.im.src];
    const int nblk = x.im.count / DSN_BLK, b0 = x.im.dst / DSN_BLK;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nblk * 2048; e += gridDim.x * blockDim.x)
        dst16[dsn_stream16_index(b0 + e / 2048, e % 2048)] = dsn_pack_value16(x, src, e / 2048, e % 2048);
}
#define DSN_NUM_STREAM_IMAGES 17

// host twin of k_pack_params (same dsn_pack_value): lets tests check the MFMA operand layout on a CPU
void dsn_pack_params_host(const float* const* params33_host, float* packed_host) {
    for (int i = 0; i < DSN_NUM_IMAGES; ++i) {
        const DsnImageX x = h_images[i];
        const float* src = params33_host[x.im.src];
        for (int e = 0; e < x.im.count; ++e) packed_host[x.im.dst + e] = dsn_pack_value(x, src, e);
    }
    _Float16* d16 = reinterpret_cast<_Float16*>(packed_host + OFF16_BASE);
    for (int i = 0; i < DSN_NUM_STREAM_IMAGES; ++i) {
        const DsnImageX x = h_images[i];
        const float* src = params33_host[x.im.src];
        const int nblk = x.im.count / DSN_BLK, b0 = x.im.dst / DSN_BLK;
        for (int e = 0; e < nblk * 2048; ++e) d16[dsn_stream16_index(b0 + e / 2048, e % 2048)] = dsn_pack_value16(x, src, e / 2048, e % 2048);
    }
}

void dsn_launch_pack_params(const float* const* params33, float* packed, hipStream_t st) {
    DsnParamPtrs pp;
    for (int i = 0; i < DSN_NUM_PARAMS_INTERNAL; ++i) pp.p[i] = params33[i];
    hipLaunchKernelGGL(k_pack_params, dim3(64, DSN_NUM_IMAGES), dim3(256), 0, st, pp, packed);
    hipLaunchKernelGGL(k_pack_params16, dim3(64, DSN_NUM_STREAM_IMAGES), dim3(256), 0, st, pp,
                       reinterpret_cast<_Float16*>(packed + OFF16_BASE));
}

// ---------------------------------------------------------------------------------------------
// building blocks
// ---------------------------------------------------------------------------------------------
// The A-operand images are laid out in HBM in exactly the order the kernel consumes them
// (OFF_L0 ... OFF_L0T, dsn_common.h), so the whole network is ONE linear weight stream per wave.
// DsnWStream keeps the current 4 KB block (16 k-steps x 64 lanes) in 4 float4 registers and issues the
// loads of the NEXT block before the 16 MFMAs of the current one: the ~1 us of matrix work per block
// covers the L2 latency of the prefetch (hipcc does not software-pipeline this on its own).
struct DsnWStream {
    const float4* p;   // this lane's slot in the current block
    float4 c0, c1, c2, c3;
    __device__ __forceinline__ void seek(const float* base, int lane) {
        p = reinterpret_cast<const float4*>(base) + lane;
        c0 = p[0]; c1 = p[64]; c2 = p[128]; c3 = p[192];
    }
};

// acc += W[32 rows of block][32*KB k] * in, KB blocks of 16 k-steps, consuming KB blocks of the stream.
template <int KB>
__device__ __forceinline__ f32x16 dsn_dense(DsnWStream& ws, const f32x16 (&in)[KB], f32x16 acc) {
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        const float4* np = ws.p + 256;   // next block (+4096 B)
        const float4 n0 = np[0], n1 = np[64], n2 = np[128], n3 = np[192];
        acc = DSN_MFMA(ws.c0.x, in[kb][0], acc);  acc = DSN_MFMA(ws.c0.y, in[kb][1], acc);
        acc = DSN_MFMA(ws.c0.z, in[kb][2], acc);  acc = DSN_MFMA(ws.c0.w, in[kb][3], acc);
        acc = DSN_MFMA(ws.c1.x, in[kb][4], acc);  acc = DSN_MFMA(ws.c1.y, in[kb][5], acc);
        acc = DSN_MFMA(ws.c1.z, in[kb][6], acc);  acc = DSN_MFMA(ws.c1.w, in[kb][7], acc);
        acc = DSN_MFMA(ws.c2.x, in[kb][8], acc);  acc = DSN_MFMA(ws.c2.y, in[kb][9], acc);
        acc = DSN_MFMA(ws.c2.z, in[kb][10], acc); acc = DSN_MFMA(ws.c2.w, in[kb][11], acc);
        acc = DSN_MFMA(ws.c3.x, in[kb][12], acc); acc = DSN_MFMA(ws.c3.y, in[kb][13], acc);
        acc = DSN_MFMA(ws.c3.z, in[kb][14], acc); acc = DSN_MFMA(ws.c3.w, in[kb][15], acc);
        ws.p = np; ws.c0 = n0; ws.c1 = n1; ws.c2 = n2; ws.c3 = n3;
    }
    return acc;
}

// 16 values of a plain [rows] vector in accumulator order for block m: rows 32m + 8q + 4*half + (0..3)
__device__ __forceinline__ f32x16 dsn_load_rows(const float* __restrict__ v, int m, int half) {
    f32x16 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 t = *reinterpret_cast<const float4*>(v + 32 * m + 8 * q + 4 * half);
        o[4 * q + 0] = t.x; o[4 * q + 1] = t.y; o[4 * q + 2] = t.z; o[4 * q + 3] = t.w;
    }
    return o;
}

__device__ __forceinline__ uint32_t dsn_relu_mask(f32x16& a) {
    uint32_t m = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const bool pos = a[r] > 0.0f;
        m |= pos ? (1u << r) : 0u;
        a[r] = pos ? a[r] : 0.0f;
    }
    return m;
}
__device__ __forceinline__ void dsn_apply_mask(f32x16& a, uint32_t m) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = ((m >> r) & 1u) ? a[r] : 0.0f;
}

// one 256 -> 256 trunk layer, forward: out = relu(W in + b); returns the relu bit masks (4 words)
__device__ __forceinline__ void dsn_layer_fwd(DsnWStream& ws, const float* __restrict__ bias,
                                              const f32x16 (&in)[8], f32x16 (&out)[8], uint32_t (&mk)[4], int lane) {
    const int half = lane >> 5;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        f32x16 acc = dsn_load_rows(bias, m, half);
        acc = dsn_dense<8>(ws, in, acc);
        const uint32_t bits = dsn_relu_mask(acc);
        if (m & 1) mk[m >> 1] |= bits << 16; else mk[m >> 1] = bits;
        out[m] = acc;
    }
}
// one 256 -> 256 trunk layer, reverse: out = (W^T in) masked by the relu pattern of the layer below
__device__ __forceinline__ void dsn_layer_bwd(DsnWStream& ws, const f32x16 (&in)[8], f32x16 (&out)[8],
                                              const uint32_t (&mk)[4]) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        acc = dsn_dense<8>(ws, in, acc);
        dsn_apply_mask(acc, (mk[m >> 1] >> (16 * (m & 1))) & 0xffffu);
        out[m] = acc;
    }
}

// ---------------------------------------------------------------------------------------------
// k_field
// ---------------------------------------------------------------------------------------------
#define FIELD_THREADS 256
#define FIELD_PTS_PER_BLOCK 128

// one wave's 32 points starting at list slot `slot0` (< count, wave-uniform)
template <bool fix_nan>
__device__ __forceinline__ void
k_field_wave(const float* __restrict__ packed, const DsnFrameState* __restrict__ fs, const float* __restrict__ x_c,
             const int32_t* __restrict__ active_list, int64_t count, int64_t slot0,
             float* __restrict__ sigma, float* __restrict__ essence, float* __restrict__ grad) {
    // fix_nan != 0: the range fallback of the split-fp16 kernels (dsn_field16.hip).  A sample whose activations left the
    // fp16 range there carries sigma = NaN; this launch re-evaluates exactly those samples (waves without one move on at
    // once, the other lanes of a wave with one compute but do not write) - sigma, essence and d sigma/dx in exact fp32.
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5;
    int64_t slot = slot0 + (lane & 31);
    bool valid = slot < count;
    if (!valid) slot = count - 1;
    const int64_t pt = active_list ? (int64_t)active_list[slot] : slot;
    if (fix_nan) {
        const float s = sigma[pt];
        valid = valid && (s != s);
        if (!__any(valid)) return;   // wave-uniform
    }
    const float x0 = x_c[3 * pt], x1 = x_c[3 * pt + 1], x2 = x_c[3 * pt + 2];

    // ---- positional encoding as 32 k-steps (2 register blocks): low lanes sin / x / z, high lanes cos / y / 0
    f32x16 pe[2];
    {
        const float xa[3] = {x0, x1, x2};
#pragma unroll
        for (int t = 0; t < 30; ++t) {
            const int j = t / 3, a = t % 3;
            float s, c;
            sincosf(xa[a] * (float)(1 << j), &s, &c);
            pe[t >> 4][t & 15] = half ? c : s;
        }
        pe[1][14] = half ? x1 : x0;
        pe[1][15] = half ? 0.0f : x2;
    }

    uint32_t mk[7][4];   // relu masks of the 7 trunk layers (static indices only)
    f32x16 hA[8], hB[8];
    DsnWStream ws;
    ws.seek(packed + OFF_L0, lane);

    // ---- stage1.0 : [pe] -> 256, bias with code / pose columns folded (per frame)
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        f32x16 acc = dsn_load_rows(fs->bias0, m, half);
        acc = dsn_dense<2>(ws, pe, acc);
        const uint32_t bits = dsn_relu_mask(acc);
        if (m & 1) mk[0][m >> 1] |= bits << 16; else mk[0][m >> 1] = bits;
        hA[m] = acc;
    }
    // ---- stage1.2 / .4 / .6
    dsn_layer_fwd(ws, packed + OFF_B1 + 0 * 256, hA, hB, mk[1], lane);
    dsn_layer_fwd(ws, packed + OFF_B1 + 1 * 256, hB, hA, mk[2], lane);
    dsn_layer_fwd(ws, packed + OFF_B1 + 2 * 256, hA, hB, mk[3], lane);
    // ---- stage2.0 : [h, pe] -> 256
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        f32x16 acc = dsn_load_rows(packed + OFF_B1 + 3 * 256, m, half);
        acc = dsn_dense<8>(ws, hB, acc);
        acc = dsn_dense<2>(ws, pe, acc);
        const uint32_t bits = dsn_relu_mask(acc);
        if (m & 1) mk[4][m >> 1] |= bits << 16; else mk[4][m >> 1] = bits;
        hA[m] = acc;
    }
    dsn_layer_fwd(ws, packed + OFF_B1 + 4 * 256, hA, hB, mk[5], lane);
    dsn_layer_fwd(ws, packed + OFF_B1 + 5 * 256, hB, hA, mk[6], lane);

    // ---- heads.  density_net: 256 -> 1 (no activation) as a per-lane dot + cross-half add.
    {
        float part = 0.0f;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const f32x16 wd = dsn_load_rows(packed + OFF_WDEN, m, half);
#pragma unroll
            for (int r = 0; r < 16; ++r) part = fmaf(wd[r], hA[m][r], part);
        }
        part += __shfl_xor(part, 32);
        const float sg = part + packed[OFF_SCAL + 0];
        if (valid && half == 0) sigma[pt] = sg;
    }
    if (essence) {   // rgb_net: relu (no-op) -> 256 -> 128 -> relu -> 3
        float e0 = 0.f, e1 = 0.f, e2 = 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            f32x16 acc = dsn_load_rows(packed + OFF_BRGB1, m, half);
            acc = dsn_dense<8>(ws, hA, acc);
            const f32x16 w0 = dsn_load_rows(packed + OFF_WRGB3 + 0 * 128, m, half);
            const f32x16 w1 = dsn_load_rows(packed + OFF_WRGB3 + 1 * 128, m, half);
            const f32x16 w2 = dsn_load_rows(packed + OFF_WRGB3 + 2 * 128, m, half);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[r] > 0.0f ? acc[r] : 0.0f;
                e0 = fmaf(w0[r], v, e0); e1 = fmaf(w1[r], v, e1); e2 = fmaf(w2[r], v, e2);
            }
        }
        e0 += __shfl_xor(e0, 32); e1 += __shfl_xor(e1, 32); e2 += __shfl_xor(e2, 32);
        if (valid && half == 0) {
            essence[3 * pt + 0] = e0 + packed[OFF_SCAL + 1];
            essence[3 * pt + 1] = e1 + packed[OFF_SCAL + 2];
            essence[3 * pt + 2] = e2 + packed[OFF_SCAL + 3];
        }
    }
    if (!grad) return;
    if (!essence) ws.seek(packed + OFF_L6T, lane);   // density-only callers skip the colour head's blocks

    // ---- reverse pass: g(h6) = W_den masked by relu(stage2.4)
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        f32x16 g = dsn_load_rows(packed + OFF_WDEN, m, half);
        dsn_apply_mask(g, (mk[6][m >> 1] >> (16 * (m & 1))) & 0xffffu);
        hA[m] = g;
    }
    dsn_layer_bwd(ws, hA, hB, mk[5]);
    dsn_layer_bwd(ws, hB, hA, mk[4]);
    // stage2.0^T : 256 -> [256 h | 64 pe]
    f32x16 dpe[2];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        acc = dsn_dense<8>(ws, hA, acc);
        dsn_apply_mask(acc, (mk[3][m >> 1] >> (16 * (m & 1))) & 0xffffu);
        hB[m] = acc;
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        dpe[b] = dsn_dense<8>(ws, hA, acc);
    }
    dsn_layer_bwd(ws, hB, hA, mk[2]);
    dsn_layer_bwd(ws, hA, hB, mk[1]);
    dsn_layer_bwd(ws, hB, hA, mk[0]);
    // stage1.0^T : 256 -> 64 pe, accumulated on top of the skip-connection part
#pragma unroll
    for (int b = 0; b < 2; ++b) dpe[b] = dsn_dense<8>(ws, hA, dpe[b]);

    // ---- encoding backward: d/dx sin(2^j x) = 2^j cos(2^j x), d/dx cos(2^j x) = -2^j sin(2^j x).
    // The partner value (cos for the sin lanes and vice versa) sits in the other half of the wave.
    {
        float g[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 30; ++t) {
            const int j = t / 3, a = t % 3;
            const float partner = __shfl_xor(pe[t >> 4][t & 15], 32);
            const float d = dpe[t >> 4][t & 15] * partner;
            const float term = d * (float)(1 << j);
            g[a] += half ? -term : term;
        }
        // identity columns: low lanes hold d/dx (t=30) and d/dz (t=31), high lanes d/dy (t=30)
        const float i30 = dpe[1][14], i31 = dpe[1][15];
        if (half) g[1] += i30; else { g[0] += i30; g[2] += i31; }
        g[0] += __shfl_xor(g[0], 32); g[1] += __shfl_xor(g[1], 32); g[2] += __shfl_xor(g[2], 32);
        if (valid && half == 0) { grad[3 * pt] = g[0]; grad[3 * pt + 1] = g[1]; grad[3 * pt + 2] = g[2]; }
    }
}

template <bool FIX>
__global__ void __launch_bounds__(FIELD_THREADS, 1)
k_field(const float* __restrict__ packed, const DsnFrameState* __restrict__ fs, const float* __restrict__ x_c,
        int64_t N, const int32_t* __restrict__ active_list, const int32_t* __restrict__ active_count,
        float* __restrict__ sigma, float* __restrict__ essence, float* __restrict__ grad, const int32_t* __restrict__ flag_count) {
    DSN_OWN_SIMD();      // (480 registers as it is: 32 would be left for another kernel's waves - see dsn_common.h)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // (fix) the split-fp16 launches count what they flag: nothing flagged - the normal case - and there is nothing to look for
    if (FIX && flag_count && *flag_count == 0) return;
    const int64_t count = active_list ? (int64_t)(*active_count) : N;
    if (!FIX) {      // (its own instantiation: the loop below costs the plain kernel 6 % through register allocation)
        const int64_t slot0 = ((int64_t)blockIdx.x * 4 + wave) * 32;
        if (slot0 >= count) return;   // wave-uniform
        k_field_wave<false>(packed, fs, x_c, active_list, count, slot0, sigma, essence, grad);
        return;
    }
    // normal launches: one 128-point tile per workgroup.  The range fallback (fix_nan) is launched with a small grid that
    // strides over the list: it normally finds nothing, and 131 072 workgroups that only look cost 0.14 ms per frame
#pragma nounroll
    for (int64_t blk = blockIdx.x;; blk += gridDim.x) {
        const int64_t slot0 = (blk * 4 + wave) * 32;
        if (slot0 >= count) return;   // wave-uniform
        // (the pointers are laundered through an empty asm: otherwise the loop-invariant loads of biases and head vectors
        //  are hoisted out of the loop and the kernel spills thousands of registers)
        const float* pk = packed;
        const DsnFrameState* f = fs;
        asm volatile("" : "+s"(pk), "+s"(f));
        k_field_wave<true>(pk, f, x_c, active_list, count, slot0, sigma, essence, grad);
    }
}

void dsn_launch_field(const float* packed, const DsnFrameState* fs, const float* x_c, int64_t N,
                      const int32_t* active_list, const int32_t* active_count, float* sigma, float* essence,
                      float* grad, hipStream_t st) {
    int64_t blocks = (N + FIELD_PTS_PER_BLOCK - 1) / FIELD_PTS_PER_BLOCK;
    if (blocks == 0) return;
    hipLaunchKernelGGL(k_field<false>, dim3((unsigned)blocks), dim3(FIELD_THREADS), 0, st, packed, fs, x_c, N, active_list,
                       active_count, sigma, essence, grad, (const int32_t*)nullptr);
}
// range fallback of the split-fp16 kernels: exact-fp32 re-evaluation of the listed samples whose sigma is NaN
void dsn_launch_field_fix(const float* packed, const DsnFrameState* fs, const float* x_c, int64_t N,
                          const int32_t* active_list, const int32_t* active_count, float* sigma, float* essence,
                          float* grad, hipStream_t st, const int32_t* flag_count) {
    int64_t blocks = (N + FIELD_PTS_PER_BLOCK - 1) / FIELD_PTS_PER_BLOCK;
    if (blocks == 0) return;
    // grid-stride (see k_field); one workgroup per compute unit: its waves own their SIMDs (DSN_OWN_SIMD), so more of them would only
    // queue up to find, in the normal case, that nothing was flagged (2048 of them: 0.04 ms per frame)
    if (blocks > dsn_cu_count_raw()) blocks = dsn_cu_count_raw();
    hipLaunchKernelGGL(k_field<true>, dim3((unsigned)blocks), dim3(FIELD_THREADS), 0, st, packed, fs, x_c, N, active_list,
                       active_count, sigma, essence, grad, flag_count);
}

// ---------------------------------------------------------------------------------------------
// k_light : model/spacenet.py:254-265 (rotation / light-centre edits) + :174-188 LightingMLP
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(FIELD_THREADS, 1)
k_light(const float* __restrict__ packed, const DsnFrameState* __restrict__ fs, const float* __restrict__ n_w,
        const float* __restrict__ x_w_pts, const float* __restrict__ ray_o, const float* __restrict__ ray_d,
        const float* __restrict__ z_vals, const float* essence, int64_t N, int S,
        const int32_t* __restrict__ active_list, const int32_t* __restrict__ active_count,
        float* colour) {      // (essence and colour may be the same array: see k_light16)
    // (round 6: the exact-fp32 twin of k_light16 was the one matrix kernel nobody had measured beside other streams'
    //  kernels - tests/test_guard_coverage.py; it is the calibration / fallback path, so it simply takes the guard)
    DSN_OWN_SIMD();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5;
    const int64_t count = active_list ? (int64_t)(*active_count) : N;
    const int64_t slot0 = ((int64_t)blockIdx.x * 4 + wave) * 32;
    if (slot0 >= count) return;
    int64_t slot = slot0 + (lane & 31);
    const bool valid = slot < count;
    if (!valid) slot = count - 1;
    const int64_t pt = active_list ? (int64_t)active_list[slot] : slot;
    const int64_t ray = pt / S;

    float in9[10];
    in9[0] = n_w[3 * pt]; in9[1] = n_w[3 * pt + 1]; in9[2] = n_w[3 * pt + 2];
    float xw[3];
    const float d[3] = {ray_d[3 * ray], ray_d[3 * ray + 1], ray_d[3 * ray + 2]};
    if (x_w_pts) { xw[0] = x_w_pts[3 * pt]; xw[1] = x_w_pts[3 * pt + 1]; xw[2] = x_w_pts[3 * pt + 2]; }
    else {
        const float z = z_vals[pt];
        xw[0] = ray_o[3 * ray] + d[0] * z; xw[1] = ray_o[3 * ray + 1] + d[1] * z; xw[2] = ray_o[3 * ray + 2] + d[2] * z;
    }
    if (fs->has_rot != 0.0f) {
        const float ax = xw[0] - fs->rot_center[0], ay = xw[1] - fs->rot_center[1];
        const float nx = (ax * fs->rot[0] + ay * fs->rot[2]) + fs->rot_center[0];
        const float ny = (ax * fs->rot[1] + ay * fs->rot[3]) + fs->rot_center[1];
        xw[0] = nx; xw[1] = ny;
    }
    if (fs->has_light != 0.0f) { xw[0] += fs->light_shift[0]; xw[1] += fs->light_shift[1]; xw[2] += fs->light_shift[2]; }
    in9[3] = xw[0]; in9[4] = xw[1]; in9[5] = xw[2];
    const float vn = dsn_norm3(d);
    in9[6] = dsn_div(d[0], vn); in9[7] = dsn_div(d[1], vn); in9[8] = dsn_div(d[2], vn);
    in9[9] = 0.0f;

    // layer 0: 9 -> 128 (5 k-steps: low lanes feature 2s, high lanes 2s+1)
    f32x16 h1[4], h2[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        f32x16 acc = dsn_load_rows(packed + OFF_BLT0, m, half);
        const float4* wp = reinterpret_cast<const float4*>(packed + OFF_LT0 + m * DSN_BLK) + lane;
        const float4 w0 = wp[0], w1 = wp[64];
        acc = DSN_MFMA(w0.x, half ? in9[1] : in9[0], acc);
        acc = DSN_MFMA(w0.y, half ? in9[3] : in9[2], acc);
        acc = DSN_MFMA(w0.z, half ? in9[5] : in9[4], acc);
        acc = DSN_MFMA(w0.w, half ? in9[7] : in9[6], acc);
        acc = DSN_MFMA(w1.x, half ? in9[9] : in9[8], acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = acc[r] > 0.0f ? acc[r] : 0.0f;
        h1[m] = acc;
    }
    DsnWStream ws;
    ws.seek(packed + OFF_LT1, lane);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        f32x16 acc = dsn_load_rows(packed + OFF_BLT1, m, half);
        acc = dsn_dense<4>(ws, h1, acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = acc[r] > 0.0f ? acc[r] : 0.0f;
        h2[m] = acc;
    }
    float part = 0.0f;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const f32x16 w = dsn_load_rows(packed + OFF_WLT2, m, half);
#pragma unroll
        for (int r = 0; r < 16; ++r) part = fmaf(w[r], h2[m][r], part);
    }
    part += __shfl_xor(part, 32);
    const float o = part + packed[OFF_SCAL + 4];
    const float wgt = (o > 0.0f ? o : expm1f(o)) + 1.0f;   // ELU(alpha=1) + 1
    if (valid && half == 0) {
        colour[3 * pt + 0] = wgt * essence[3 * pt + 0];
        colour[3 * pt + 1] = wgt * essence[3 * pt + 1];
        colour[3 * pt + 2] = wgt * essence[3 * pt + 2];
    }
}

void dsn_launch_light(const float* packed, const DsnFrameState* fs, const float* n_w, const float* x_w,
                      const float* ray_o, const float* ray_d, const float* z_vals, const float* essence, int64_t N,
                      int S, const int32_t* active_list, const int32_t* active_count, float* colour, hipStream_t st) {
    int64_t blocks = (N + FIELD_PTS_PER_BLOCK - 1) / FIELD_PTS_PER_BLOCK;
    if (blocks == 0) return;
    hipLaunchKernelGGL(k_light, dim3((unsigned)blocks), dim3(FIELD_THREADS), 0, st, packed, fs, n_w, x_w, ray_o, ray_d,
                       z_vals, essence, N, S, active_list, active_count, colour);
}
