// dsn_field16.hip - k_field16: the canonical field + d sigma/dx on the gfx950 matrix cores at 16x the
// fp32-MFMA issue rate, with fp32-equivalent accuracy ("split-fp16", 3 products).
//
// Numerics.  Every operand v (activation or weight) is split  v = hi + lo * 2^-12  with
//   hi = fp16(v),  lo = fp16((v - hi) * 2^12)      (22 significand bits kept)
// and  W.x  is evaluated as   (W_hi x_hi)  +  2^-12 (W_hi x_lo + W_lo x_hi)   with
// v_mfma_f32_32x32x16_f16 (fp32 accumulate; products of fp16 are exact in fp32).  The dropped
// W_lo x_lo term is 2^-24 relative - the size of one fp32 rounding.  Measured against the float64
// reference this is as accurate as the reference's own float32 run (tests/test_gpu_stages.py::test_field;
// sigma 8e-6 abs).  Plain bf16 / fp16 inputs miss the 1e-4 bar by 10x, bf16 2-way split by 1.4x.
// Range: the reverse pass runs on g * 2^-6 (exact rescale at the end) so that |g| up to 4e6 stays
// inside fp16; forward activations must stay below 65504 (they are O(10) for NeRF trunks).  Every epilogue keeps a
// running maximum of the values it splits (one v_max3 per two elements); a sample that reaches F16_RANGE anywhere is
// FLAGGED: its sigma is written as NaN and the exact-fp32 kernel re-evaluates it (dsn_launch_field_fix, dsn_field.hip) -
// checkpoints whose activations or adjoints leave the fp16 range render correctly, only slower (tests/test_gpu_round2.py: test_fp16_range_fallback_*).
//
// Structure.  Same transposed formulation and register chaining as k_field (dsn_field.hip): one
// wavefront owns 32 points, the accumulator layout of the 32x32 MFMA is re-used as the next
// B operand - here after an in-register fp32 -> (hi, lo) fp16 pack, 8 k-values per lane per step.
// 3 MFMAs of 32 cycles replace 8 fp32 MFMAs of 64 cycles: 167 424 matrix cycles per 32 points
// instead of 884 736.  At that rate the weight stream (3.57 MB per 128 points per CU) can no longer be
// fetched per wave from L2, so the 4 waves of a workgroup share it: the stream is cut into 4 KB
// blocks, a 16-slot LDS ring (64 KB) is filled 8 blocks ahead with global_load_lds_dwordx4 (no VGPR
// staging; each wave moves one 1 KB quarter of every block), one barrier per 8 blocks.
#include "dsn_common.h"
#include "dsn_kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

// build-time variant switches (A/B-ed with scripts/variants.sh + scripts/gpu_variants.sh; defaults = best measured).
// Every switch below keeps the results.  The timing ablations that do NOT (F16_ABL, F16_SABL: kernels with pieces cut out, WRONG
// results by design) and the in-kernel cycle counters (F16_TIMING) exist only in builds made with -DDSN_EXPERIMENTS
// (scripts/variants.sh passes it): the product build (dual-space-nerf_amd/build.py) cannot switch them on.
#if !defined(DSN_EXPERIMENTS) && (defined(F16_ABL) || defined(F16_SABL) || defined(F16_TIMING) || defined(F16_TRAIN_ABL))
#error "F16_ABL / F16_SABL / F16_TIMING are experiment switches (wrong results / debug counters): build with -DDSN_EXPERIMENTS"
#endif
#ifndef F16_SINCOS_OCML
#define F16_SINCOS_OCML 0     // 1: ocml sincosf (divergent large-argument path), 0: branch-free dsn_sincos
#endif
#ifndef F16_FENCE
#define F16_FENCE 0           // sched_barrier after every block
#endif
#ifndef F16_SGB
#define F16_SGB 1             // sched_group_barrier interleave (1 MFMA : 4 VALU) inside every block + fence per block
#endif
#ifndef F16_SGB_VALU
#define F16_SGB_VALU 4        // VALU instructions per MFMA in that pattern
#endif
#ifndef F16_SGB_DS
#define F16_SGB_DS 0          // 1: pin one operand ds_read behind each of the first four MFMAs of a block
#endif
#ifndef F16_MIX
#define F16_MIX 1             // hi / lo split of the pipelined epilogues with v_cvt_pk_f16_f32 + v_fma_mix{lo,hi}_f16 (inline asm)
#endif
#ifndef F16_PIPE_LATE
#define F16_PIPE_LATE 1       // stage2.0 / stage2.4 (forward) with software-pipelined epilogues like the plain 256 -> 256 layers
#endif
#ifndef F16_ABL
#define F16_ABL 0             // timing ablations (WRONG results): 1 no ring barrier, 2 no vmcnt wait, 4 no LDS-DMA, 8 no pipelined epilogue (MFMAs die too), 16 empty epilogue slices (MFMAs kept), 32 epilogue without the second accumulator's add / fold
#endif
#ifndef F16_NEXT_POINT
#define F16_NEXT_POINT 1      // k_field16: the next tile's list entry and coordinates are fetched under the current tile
#endif
#ifndef F16_DMA_SPREAD
#define F16_DMA_SPREAD 1      // LDS-DMA of the next chunk two pieces per block over four blocks instead of eight pieces behind the barrier
#endif
#ifndef F16_BIAS_AHEAD
#define F16_BIAS_AHEAD 1      // forward layers: read the next output block's bias rows one block early
#endif
#ifndef F16_TIMING
#define F16_TIMING 0          // 1 (debug variant): phase cycle counters of k_field16<forward>, read by dsn_debug_timing
#endif
#ifndef F16_PREFETCH
#define F16_PREFETCH 1        // explicit one-block-ahead LDS operand reads
#endif

#define F16_THREADS 256
// offset of output block m inside a training activation row ([N,256] row-major: 32 m).  Experiment builds (F16_TRAIN_ABL & 32, timing only,
// WRONG addresses for the readers) emulate a TILE-major layout instead: [tile of 32 list slots][m][row][32 features] - 4 KB contiguous
// per wave and output block
#if defined(DSN_EXPERIMENTS) && defined(F16_TRAIN_ABL) && (F16_TRAIN_ABL & 32)
#define F16_ST_M(m) (1024 * (m))
#else
#define F16_ST_M(m) (32 * (m))
#endif
#define F16_CHUNK 8              // blocks per ring barrier; the ring holds two chunks (fixed: the DMA immediates span one chunk)
#define F16_RING_SLOTS (2 * F16_CHUNK)
#define F16_NCHUNK (DSN_STREAM_BLOCKS / F16_CHUNK)   // 109
#define F16_RANGE 65000.0f                           // |value| an epilogue may hand to the fp16 split (fp16 max 65504)
#define F16_GSCALE 0.015625f                         // reverse pass runs on g / 64
#define F16_GUNSCALE 64.0f

#if F16_TIMING
__device__ unsigned long long g_f16_timing[16];
#define F16_STAMP(i) do { if (MODE == F16_FWD) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); tstamp[i] = __builtin_readcyclecounter(); rstamp[i] = wall_clock64(); } } while (0)
extern "C" __attribute__((visibility("default"))) int dsn_debug_timing(unsigned long long* out16, int reset) {
    if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_f16_timing), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_f16_timing), z, sizeof(z)) != hipSuccess) return 1; }
    return 0;
}
#else
#define F16_STAMP(i) do { } while (0)
#endif
struct W16 {                 // weight stream state of one wave
    const char* g;           // this lane's source: stream base + wave * 8192 + 4096 + lane * 16 (chunk-major layout, dsn_stream16_index)
    char* ring;              // LDS ring base: [chunk parity][quarter][block in chunk][1 KB]
    unsigned ring_off;       // its LDS byte address (for M0)
    int wave;
    int nchunk = F16_NCHUNK; // chunks [.., nchunk) are what this kernel streams (the forward-only kernel stops at the reverse layers)
    bool spread = F16_DMA_SPREAD != 0;   // LDS-DMA of the next chunk two pieces per block over four blocks (false: all eight behind the ring
                                         // barrier - the kernels that also STORE activations every chunk run 2-3 % faster that way: their
                                         // stores and the spread pieces share one queue in front of the boundary's vmcnt(0))
    half8 h0, l0, h1, l1;    // the CURRENT block's operands: (hi, lo) for k-step 0 and 1
};

// LDS-DMA of chunk c: this wave moves quarter `wave` of its 8 blocks = 8 contiguous KB, in memory and in the ring, so
// ONE global address + ONE M0 + 8 immediate offsets (-4096 .. 3072, applied to both sides by the hardware) do it.
// Issued through inline asm: the compiler-visible builtin makes hipcc put s_waitcnt vmcnt(0) in front of the next ds_read
// (it cannot prove the ring slots differ), which serialises the prefetch.  M0 is declared clobbered (nothing else in these
// kernels lives in it).  Completion is waited for by w16_boundary's own vmcnt(0) + barrier.
#ifndef F16_POS_OFFS
#define F16_POS_OFFS 0        // 1 (experiment, same values): the LDS-DMA pieces with non-negative immediate offsets only (two M0 / address bases)
#endif
__device__ __forceinline__ void w16_stage(const W16& w, int c) {
    const char* src = w.g + (size_t)c * 32768;
    const unsigned dst = w.ring_off + (c & 1) * 32768 + w.wave * 8192 + 4096;
#if F16_POS_OFFS
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %0, off\n\tglobal_load_lds_dwordx4 %0, off offset:1024\n\t"
                 "global_load_lds_dwordx4 %0, off offset:2048\n\tglobal_load_lds_dwordx4 %0, off offset:3072"
                 : : "v"(src - 4096), "s"(dst - 4096) : "memory", "m0");
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %0, off\n\tglobal_load_lds_dwordx4 %0, off offset:1024\n\t"
                 "global_load_lds_dwordx4 %0, off offset:2048\n\tglobal_load_lds_dwordx4 %0, off offset:3072"
                 : : "v"(src), "s"(dst) : "memory", "m0");
    return;
#endif
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %0, off offset:-4096\n\tglobal_load_lds_dwordx4 %0, off offset:-3072\n\t"
                 "global_load_lds_dwordx4 %0, off offset:-2048\n\tglobal_load_lds_dwordx4 %0, off offset:-1024\n\t"
                 "global_load_lds_dwordx4 %0, off\n\tglobal_load_lds_dwordx4 %0, off offset:1024\n\t"
                 "global_load_lds_dwordx4 %0, off offset:2048\n\tglobal_load_lds_dwordx4 %0, off offset:3072"
                 : : "v"(src), "s"(dst) : "memory", "m0");
}
// the same LDS-DMA two pieces at a time (F16_DMA_SPREAD): all four waves issuing their eight 1 KB pieces right behind the ring
// barrier queue 32 KB on the CU's one address path while the matrix pipe runs dry (in-order issue: a wave sits in its VMEM
// instructions); two pieces behind each of the first four blocks of a chunk keep that path a quarter busy instead
template <int I>
__device__ __forceinline__ void w16_stage_part(const W16& w, int c) {
    const char* src = w.g + (size_t)c * 32768;
    const unsigned dst = w.ring_off + (c & 1) * 32768 + w.wave * 8192 + 4096;
#if F16_POS_OFFS
    {
        const int back = I < 2 ? 4096 : 0;
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %0, off offset:%2\n\tglobal_load_lds_dwordx4 %0, off offset:%3"
                     : : "v"(src - back), "s"(dst - back), "n"(2048 * (I & 1)), "n"(2048 * (I & 1) + 1024) : "memory", "m0");
        return;
    }
#endif
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %0, off offset:%2\n\tglobal_load_lds_dwordx4 %0, off offset:%3"
                 : : "v"(src), "s"(dst), "n"(-4096 + 2048 * I), "n"(-3072 + 2048 * I) : "memory", "m0");
}
// chunk boundary in front of block b (b % 8 == 0): after the barrier chunk b/8 has landed for everyone and chunk
// b/8 - 1 has been read by everyone (its last block is already in registers) -> its half of the ring takes chunk b/8 + 1
__device__ __forceinline__ void w16_boundary(const W16& w, int b) {
#if !(F16_ABL & 2)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's quarters of chunk b/8 have landed in LDS
#endif
#if !(F16_ABL & 1)
    __syncthreads();
#endif
    const int c = b / F16_CHUNK;
#if !(F16_ABL & 4)
    if (!w.spread && c + 1 < w.nchunk) w16_stage(w, c + 1);
#endif
}
__device__ __forceinline__ void w16_read(const W16& w, int b, int lane, half8& h0, half8& l0, half8& h1, half8& l1) {
    const char* s = w.ring + ((b >> 3) & 1) * 32768 + (b & 7) * 1024 + lane * 16;
    h0 = *reinterpret_cast<const half8*>(s);
    l0 = *reinterpret_cast<const half8*>(s + 8192);
    h1 = *reinterpret_cast<const half8*>(s + 16384);
    l1 = *reinterpret_cast<const half8*>(s + 24576);
}
// w16_begin = w16_begin_issue (the first chunk's LDS-DMA: as early in the kernel as possible, its L2 round trip then runs under the
// rest of the prologue) + w16_begin_wait
__device__ __forceinline__ void w16_begin_issue(W16& w, int first_blk) { w16_stage(w, first_blk / F16_CHUNK); }
__device__ __forceinline__ void w16_begin_wait(W16& w, int lane, int first_blk) {
    w16_boundary(w, first_blk);
#if !(F16_ABL & 4)
    if (w.spread && first_blk / F16_CHUNK + 1 < w.nchunk) w16_stage_part<0>(w, first_blk / F16_CHUNK + 1);   // (dense16 issues parts 1..3 and every later chunk)
#endif
    w16_read(w, first_blk, lane, w.h0, w.l0, w.h1, w.l1);
}
__device__ __forceinline__ void w16_begin(W16& w, int lane, int first_blk) {
    w16_begin_issue(w, first_blk);
    w16_begin_wait(w, lane, first_blk);
}

// (accM, accC) += W[32 rows][32*KB k] * (xh, xl): consumes KB blocks starting at stream block `blk`.
// One-block-ahead software pipeline: the LDS reads of block b+1 are issued before the 6 MFMAs of block b.
struct NoHook { __device__ __forceinline__ void operator()(int) const {} };

// XLS = true : reverse pass - xl and Wl carry a 2^12 scale    -> accM += Wh xh,  accC += Wh xl + Wl xh
// XLS = false: forward pass - the weight images are pre-scaled by 2^6 as a whole (hi and lo alike) and xl is the
//              plain fp16 residual, so all three products share one scale and may go to either accumulator:
//              accM + accC = 64 (Wh xh + Wh xl + Wl xh)
template <int KB, bool XLS, class Hook = NoHook>
__device__ __forceinline__ void dense16(W16& w, int& blk, int lane, const half8 (&xh)[KB][2], const half8 (&xl)[KB][2],
                                        f32x16& accM, f32x16& accC, Hook&& hook = NoHook()) {
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
#if F16_PREFETCH
        half8 n0 = w.h0, m0 = w.l0, n1 = w.h1, m1 = w.l1;
        if (blk + 1 < DSN_STREAM_BLOCKS) {
            if (((blk + 1) & (F16_CHUNK - 1)) == 0) w16_boundary(w, blk + 1);
#if !(F16_ABL & 4)
            if (w.spread) {   // pieces of the chunk after the one block blk + 1 lives in, behind its first four blocks
                const int pos = (blk + 1) & (F16_CHUNK - 1), cn = (blk + 1) / F16_CHUNK + 1;
                if (cn < w.nchunk) {
                    if (pos == 0) w16_stage_part<0>(w, cn);
                    if (pos == 1) w16_stage_part<1>(w, cn);
                    if (pos == 2) w16_stage_part<2>(w, cn);
                    if (pos == 3) w16_stage_part<3>(w, cn);
                }
            }
#endif
            w16_read(w, blk + 1, lane, n0, m0, n1, m1);
        }
#else
        if (blk > 0) {
            if ((blk & (F16_CHUNK - 1)) == 0) w16_boundary(w, blk);
            w16_read(w, blk, lane, w.h0, w.l0, w.h1, w.l1);
        }
#endif
        if (XLS) {
            accM = MFMA16(w.h0, xh[kb][0], accM);
            accC = MFMA16(w.h0, xl[kb][0], accC);
            accC = MFMA16(w.l0, xh[kb][0], accC);
            accM = MFMA16(w.h1, xh[kb][1], accM);
            accC = MFMA16(w.h1, xl[kb][1], accC);
            accC = MFMA16(w.l1, xh[kb][1], accC);
        } else {
            // strictly alternating accumulators: a dependent MFMA issued right behind VALU fillers waits for the full
            // write-back of its predecessor (MI355X_MICROARCH.md: +43 cycles), an independent one does not
            accM = MFMA16(w.h0, xh[kb][0], accM);
            accC = MFMA16(w.l0, xh[kb][0], accC);
            accM = MFMA16(w.h0, xl[kb][0], accM);
            accC = MFMA16(w.h1, xh[kb][1], accC);
            accM = MFMA16(w.l1, xh[kb][1], accM);
            accC = MFMA16(w.h1, xl[kb][1], accC);
        }
#if !(F16_ABL & 8)
        hook(kb);   // independent VALU work (the previous output block's epilogue slice) issues under these MFMAs
#endif
#if F16_PREFETCH
        w.h0 = n0; w.l0 = m0; w.h1 = n1; w.l1 = m1;
#endif
        ++blk;
#if F16_SGB
        // pin the issue order inside this block: MFMA, 4 VALU (epilogue slice of the previous output block), MFMA, ...
        // (hipcc otherwise clusters all MFMAs of a chunk and leaves the VALU work as an unoverlapped tail)
#if F16_SGB_DS == 2
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);   // the next block's four operand reads first: a full block (6 MFMAs) of lead
#endif
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#if F16_SGB_DS == 1
            if (i < 4) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // one operand ds_read behind each of 4 MFMAs
#endif
            __builtin_amdgcn_sched_group_barrier(0x002, F16_SGB_VALU, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#endif
#if F16_FENCE
        __builtin_amdgcn_sched_barrier(0);   // keep the pipeline depth at one block (bounds the operand registers)
#endif
    }
}

__device__ __forceinline__ f32x16 rows16(const float* __restrict__ v, int m, int half) {
    f32x16 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 t = *reinterpret_cast<const float4*>(v + 32 * m + 8 * q + 4 * half);
        o[4 * q + 0] = t.x; o[4 * q + 1] = t.y; o[4 * q + 2] = t.z; o[4 * q + 3] = t.w;
    }
    return o;
}
__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.0f;
    return z;
}
// fp32 block (accumulator layout) -> the two k-steps of the next B operand, split hi / lo.
// v - (float)hi is exact in fp32; written as an fma so that it maps onto v_fma_mix_f32 (fp16 source, no separate
// conversion).  SCALED: lo carries 2^12 (reverse pass: adjoints span many decades).  Unscaled (forward): the plain
// fp16 residual - its absolute error is <= max(2^-25, 2^-22 |v|), i.e. fp32 rounding of the O(1) sums it feeds.
template <bool SCALED>
__device__ __forceinline__ void split16(const f32x16& v, half8 (&h)[2], half8 (&l)[2]) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const _Float16 hi = (_Float16)v[r];
        const float res = fmaf((float)hi, -1.0f, v[r]);
        h[r >> 3][r & 7] = hi;
        l[r >> 3][r & 7] = SCALED ? (_Float16)(res * DSN_LO_SCALE) : (_Float16)res;
    }
}
// forward accumulators hold 64 z (see dense16)
#define F16_FWD_SCALE 64.0f
#define F16_FWD_INV 0.015625f
__device__ __forceinline__ f32x16 unscale16(const f32x16& m, const f32x16& c) {
    f32x16 v;
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = (m[r] + c[r]) * F16_FWD_INV;
    return v;
}
__device__ __forceinline__ f32x16 fold16(const f32x16& m, const f32x16& c) {
    f32x16 v;
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = fmaf(c[r], DSN_LO_INV, m[r]);
    return v;
}
// relu + its bit pattern / mask application
// Pattern word of 16 accumulator elements: bit (15 - r) set <=> element r is active.  Two VALU ops per element:
// v_alignbit shifts the sign bit of the pre-activation into the word, v_max applies the relu (a pre-activation of
// exactly +0 counts as active; it contributes 0 either way in the forward pass, and rounding already decides the
// pattern of any |z| < 1e-7 differently from an fp32 fma chain).
__device__ __forceinline__ uint32_t dsn_push_sign(uint32_t word, float v) {
    return __builtin_amdgcn_alignbit(word, __float_as_uint(v), 31);
}
__device__ __forceinline__ uint32_t dsn_active_word(uint32_t signs) { return ~signs & 0xffffu; }
#ifndef F16_KEEP_ASM
#define F16_KEEP_ASM 1        // relu mask of the reverse pass as v_bfe_i32 + v_and_b32 (hipcc turns the C form into v_and + v_cmp + v_cndmask through VCC)
#endif
__device__ __forceinline__ float dsn_keep_active(float v, uint32_t word, int r) {
#if F16_KEEP_ASM
    uint32_t m;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(word), "s"(15 - r));      // all ones where bit (15 - r) of the pattern word is set
    return __uint_as_float(__float_as_uint(v) & m);
#else
    return __uint_as_float(__float_as_uint(v) & (uint32_t)__builtin_amdgcn_sbfe((int)word, 15 - r, 1));
#endif
}
__device__ __forceinline__ uint32_t relu_bits16(f32x16& a) {
    uint32_t s = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        s = dsn_push_sign(s, a[r]);
        a[r] = fmaxf(a[r], 0.0f);
    }
    return dsn_active_word(s);
}
__device__ __forceinline__ void mask16(f32x16& a, uint32_t m) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = dsn_keep_active(a[r], m, r);
}

// running maximum of |v| over everything a lane splits into fp16 operands (range guard, see the header)
__device__ __forceinline__ void track16(float& ovf, const f32x16& v) {
#pragma unroll
    for (int r = 0; r < 16; r += 2) asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(ovf) : "v"(v[r]), "v"(v[r + 1]));   // (asm: see epi_slice)
}
__device__ __forceinline__ float dsn_nan_flag() { return __uint_as_float(0x7fc00000u); }

#if defined(DSN_EXPERIMENTS) && defined(F16_TRAIN_ABL) && (F16_TRAIN_ABL & 256)
// timing emulation (WRONG data, right access pattern) of WHOLE-LINE stores: instruction j writes rows 8 j .. 8 j + 7 of the wave's 32, eight
// lanes per row = the row's whole 128-byte line of this output block (instead of every lane two 16-byte pieces of its own row: 32
// partial-line requests per instruction).  The row pointers come from the rows' own lanes by two shuffles per instruction.
__device__ __forceinline__ void store16_lines_emul(float* st, const float (&o)[16]) {
    const int lane = threadIdx.x & 63;
    const unsigned long long me = (unsigned long long)(uintptr_t)st;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int src = 8 * j + (lane >> 3);      // the half-0 lane of row 8 j + lane / 8
        const unsigned lo = __shfl((unsigned)me, src), hi = __shfl((unsigned)(me >> 32), src);
        float* pj = reinterpret_cast<float*>((uintptr_t)(((unsigned long long)hi << 32) | lo));
        if (pj) *reinterpret_cast<float4*>(pj + 4 * (lane & 7)) = make_float4(o[4 * j], o[4 * j + 1], o[4 * j + 2], o[4 * j + 3]);
    }
}
#endif
// Deferred epilogue, two accumulator registers at a time.  With one wave per SIMD nothing else can cover the VALU
// work of an epilogue (fold, relu / mask, hi-lo split, pack: ~12 instructions per element), so the epilogue of output
// block m-1 is cut into 8 slices of two elements and slice kb is issued right behind the MFMAs of block (m, kb):
// the matrix pipe stays busy while the VALU retires the previous block.
// ST (training kernel only): the element values also go to `st` = this lane's row-major slot of the output block
// ([N,256] fp32 per layer: point row, features 32 m + 8 (r >> 2) + 4 half + (r & 3)), times `stscale`.
// HEAD (forward, stage2.4 only): the density head rides along - sg += w_den[feature] * relu value, in feature order
template <bool FWD, bool ST = false, bool HEAD = false>
__device__ __forceinline__ void epi_slice(const f32x16& pM, const f32x16& pC, int kb, uint32_t mword, uint32_t& bits,
                                          half8 (&yh)[2], half8 (&yl)[2], float& ovf, float* st = nullptr, float stscale = 1.0f,
                                          const float* wd = nullptr, float* sg = nullptr) {
#if F16_ABL & 16
    {   // timing ablation: the accumulators stay live (no MFMA is dead), nothing is computed from them
        asm volatile("" : : "a"(pM[2 * kb]), "a"(pM[2 * kb + 1]), "a"(pC[2 * kb]), "a"(pC[2 * kb + 1]));
        if (kb == 0) { yh[0] = yh[1]; yl[0] = yl[1]; }
        return;
    }
#endif
    float vv[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int r = 2 * kb + e;
#if F16_ABL & 32
        // timing emulation (WRONG results) of "one accumulator per output block, two blocks' chains interleaved": the epilogue loses its
        // accumulator add (forward) / fold (reverse); the second accumulator stays live so that no MFMA dies
        asm volatile("" : : "v"(pC[r]));
        float v = FWD ? pM[r] * F16_FWD_INV : pM[r];
#else
        float v = FWD ? (pM[r] + pC[r]) * F16_FWD_INV : fmaf(pC[r], DSN_LO_INV, pM[r]);
#endif
        if (FWD) {
            bits = dsn_push_sign(bits, v);      // `bits` collects the 16 signs of this output block
            v = fmaxf(v, 0.0f);
        } else {
            v = dsn_keep_active(v, mword, r);
        }
        vv[e] = v;
        if (HEAD) *sg = fmaf(wd[8 * (r >> 2) + (r & 3)], v, *sg);      // wd = this lane's 16 head weights of the block (rows16 order)
#if !F16_MIX
        const _Float16 hi = (_Float16)v;
        const float res = fmaf((float)hi, -1.0f, v);
        yh[r >> 3][r & 7] = hi;
        yl[r >> 3][r & 7] = FWD ? (_Float16)res : (_Float16)(res * DSN_LO_SCALE);
#endif
    }
#if F16_MIX
    {   // the pair's split in 3 (forward) / 5 (reverse) VALU instructions: packed convert for the hi halves, then the residuals
        // straight from the packed hi register with v_fma_mix{lo,hi}_f16 (f16 source x f32 constant + f32 addend -> f16 half of
        // the destination; hipcc emits two converts, two subtractions and a pack for the same thing)
        const int r = 2 * kb;
        uint32_t H, L;
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(H) : "v"(vv[0]), "v"(vv[1]));
        if (FWD) {
            asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(L) : "v"(H), "v"(vv[0]));
            asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(L) : "v"(H), "v"(vv[1]));
        } else {
            const float k = -DSN_LO_SCALE;
            const float s0 = vv[0] * DSN_LO_SCALE, s1 = vv[1] * DSN_LO_SCALE;      // lo = fp16((v - hi) * 2^12) = fp16(hi * -2^12 + v * 2^12)
            asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(L) : "v"(H), "s"(k), "v"(s0));
            asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(L) : "v"(H), "s"(k), "v"(s1));
        }
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        u32x4 th = __builtin_bit_cast(u32x4, yh[r >> 3]), tl = __builtin_bit_cast(u32x4, yl[r >> 3]);
        th[(r & 7) >> 1] = H;
        tl[(r & 7) >> 1] = L;
        yh[r >> 3] = __builtin_bit_cast(half8, th);
        yl[r >> 3] = __builtin_bit_cast(half8, tl);
    }
#endif
    // v_max3_f32 (relu output >= 0).  As asm: written with fmaxf the running maximum is re-associated into one tree at the end of
    // the tile inside the persistent tile loop, and every value stays live (spilled) until then
    if (FWD) asm("v_max3_f32 %0, %0, %1, %2" : "+v"(ovf) : "v"(vv[0]), "v"(vv[1]));
    else asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(ovf) : "v"(vv[0]), "v"(vv[1]));
    // Training stores.  The chunk boundary (w16_boundary: s_waitcnt vmcnt(0) for the LDS-DMA pieces) also waits for every
    // store in flight, and it sits right in front of slice 7 (blocks per output tile = blocks per chunk).  Stores issued slice
    // by slice were 1-3 blocks old at that wait and cost a full write latency per chunk (3.84 ms vs 2.34 ms without stores);
    // issued together on slice 7 they have a whole chunk to drain.  The 16 values are recomputed from the accumulators.
#if defined(DSN_EXPERIMENTS) && defined(F16_TRAIN_ABL) && (F16_TRAIN_ABL & 256)
    if (ST && kb == 7) {
        float o[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = FWD ? (pM[r] + pC[r]) * F16_FWD_INV : fmaf(pC[r], DSN_LO_INV, pM[r]);
            v = FWD ? fmaxf(v, 0.0f) : dsn_keep_active(v, mword, r);
            o[r] = v * stscale;
        }
        store16_lines_emul(st, o);
        return;
    }
#endif
    if (ST && kb == 7 && st) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * q + e;
                float v = FWD ? (pM[r] + pC[r]) * F16_FWD_INV : fmaf(pC[r], DSN_LO_INV, pM[r]);
                v = FWD ? fmaxf(v, 0.0f) : dsn_keep_active(v, mword, r);
                o[e] = v * stscale;
            }
            *reinterpret_cast<float4*>(st + 8 * q) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}
// the same store for a whole block (non-pipelined epilogues)
#ifndef F16_NT_STORE
#define F16_NT_STORE 0        // training stores (activations written once, read much later by the weight-gradient kernels) with the nt hint
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store16(float* st, const f32x16& v, float stscale) {
#if defined(DSN_EXPERIMENTS) && defined(F16_TRAIN_ABL) && (F16_TRAIN_ABL & 4)
    // timing emulation (WRONG data, right access pattern) of quad-transposed stores: instruction j writes, for the four points of a
    // lane quad in turn, point j's row - lane i of the quad its 16-byte piece i of a 64-byte segment (lower / upper half-wave: first /
    // second segment of the 128-byte line) - instead of every lane 16 bytes of its own row
    {
        const int lane = threadIdx.x & 63, hf = lane >> 5, i = lane & 3;
        const unsigned long long me = (unsigned long long)(uintptr_t)st;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int src = (lane & ~3) | j;
            const unsigned lo = __shfl((unsigned)me, src), hi = __shfl((unsigned)(me >> 32), src);
            float* pj = reinterpret_cast<float*>((uintptr_t)(((unsigned long long)hi << 32) | lo));
            if (pj) *reinterpret_cast<float4*>(pj - 4 * hf + 16 * hf + 4 * i) =
                make_float4(v[4 * j] * stscale, v[4 * j + 1] * stscale, v[4 * j + 2] * stscale, v[4 * j + 3] * stscale);
        }
        return;
    }
#endif
#if defined(DSN_EXPERIMENTS) && defined(F16_TRAIN_ABL) && (F16_TRAIN_ABL & 256)
    {
        float o[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = v[r] * stscale;
        store16_lines_emul(st, o);
        return;
    }
#endif
    if (!st) return;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#if F16_NT_STORE
        const f32x4 x = {v[4 * q] * stscale, v[4 * q + 1] * stscale, v[4 * q + 2] * stscale, v[4 * q + 3] * stscale};
        __builtin_nontemporal_store(x, reinterpret_cast<f32x4*>(st + 8 * q));
#else
        *reinterpret_cast<float4*>(st + 8 * q) = make_float4(v[4 * q] * stscale, v[4 * q + 1] * stscale, v[4 * q + 2] * stscale, v[4 * q + 3] * stscale);
#endif
    }
}

// 256 -> 256 forward layer (software-pipelined epilogues: the epilogue of output block m-1 issues under the MFMAs of block m)
//   EXTRA: stage2.0 - every output block takes two more k-blocks, the positional encoding (operands back from LDS)
//   HEAD : stage2.4 - the density head is accumulated in the epilogue (sg, feature order: bit-identical to a separate loop)
template <bool ST = false, bool EXTRA = false, bool HEAD = false>
__device__ __forceinline__ void layer16_fwd(W16& w, int& blk, int lane, const float* __restrict__ bias,
                                            const half8 (&xh)[8][2], const half8 (&xl)[8][2], half8 (&yh)[8][2],
                                            half8 (&yl)[8][2], uint32_t (&mk)[4], float& ovf, float* st = nullptr,
                                            const half8 (*s_pe)[F16_THREADS] = nullptr, const float* wden = nullptr, float* sg = nullptr) {
    const int half = lane >> 5;
    const int tid = threadIdx.x;
    f32x16 pM = zero16(), pC = zero16();
#if F16_BIAS_AHEAD
    // the accumulator's start value (64 x bias, from LDS) is read one output block ahead: read right in front of its first MFMA it
    // costs an s_waitcnt lgkmcnt(0) - a full LDS round trip with the matrix pipe idle - per output block
    f32x16 bnext = rows16(bias, 0, half);
#endif
#pragma unroll
    for (int m = 0; m < 8; ++m) {
#if F16_BIAS_AHEAD
        f32x16 aM = bnext, aC = zero16();
        if (m + 1 < 8) bnext = rows16(bias, m + 1, half);
#else
        f32x16 aM = rows16(bias, m, half), aC = zero16();
#endif
        uint32_t bits = 0;
        if (m == 0) dense16<8, false>(w, blk, lane, xh, xl, aM, aC);
        else dense16<8, false>(w, blk, lane, xh, xl, aM, aC, [&](int kb) {
            epi_slice<true, ST, HEAD>(pM, pC, kb, 0u, bits, yh[m - 1], yl[m - 1], ovf, (ST && st) ? st + F16_ST_M(m - 1) : nullptr, 1.0f,
                                      HEAD ? wden + 32 * (m - 1) + 4 * half : nullptr, sg); });
        if (EXTRA) {   // encoding operands come back from LDS just for these two blocks
            half8 qh[2][2], ql[2][2];
            qh[0][0] = s_pe[0][tid]; qh[0][1] = s_pe[1][tid]; qh[1][0] = s_pe[2][tid]; qh[1][1] = s_pe[3][tid];
            ql[0][0] = s_pe[4][tid]; ql[0][1] = s_pe[5][tid]; ql[1][0] = s_pe[6][tid]; ql[1][1] = s_pe[7][tid];
            dense16<2, false>(w, blk, lane, qh, ql, aM, aC);
        }
        if (m > 0) { if ((m - 1) & 1) mk[(m - 1) >> 1] |= dsn_active_word(bits) << 16; else mk[(m - 1) >> 1] = dsn_active_word(bits); }
        pM = aM; pC = aC;
    }
    {   // last block: nothing left to hide it under
        uint32_t bits = 0;
#pragma unroll
        for (int kb = 0; kb < 8; ++kb)
            epi_slice<true, ST, HEAD>(pM, pC, kb, 0u, bits, yh[7], yl[7], ovf, (ST && st) ? st + F16_ST_M(7) : nullptr, 1.0f,
                                      HEAD ? wden + 32 * 7 + 4 * half : nullptr, sg);
        mk[3] |= dsn_active_word(bits) << 16;
    }
}
// 256 -> 256 reverse layer
template <bool ST = false>
__device__ __forceinline__ void layer16_bwd(W16& w, int& blk, int lane, const half8 (&xh)[8][2],
                                            const half8 (&xl)[8][2], half8 (&yh)[8][2], half8 (&yl)[8][2],
                                            const uint32_t (&mk)[4], float& ovf, float* st = nullptr, float stscale = F16_GUNSCALE) {
    f32x16 pM = zero16(), pC = zero16();
    uint32_t dummy = 0;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        f32x16 aM = zero16(), aC = zero16();
        const uint32_t mw = m > 0 ? ((mk[(m - 1) >> 1] >> (16 * ((m - 1) & 1))) & 0xffffu) : 0u;
        if (m == 0) dense16<8, true>(w, blk, lane, xh, xl, aM, aC);
        else dense16<8, true>(w, blk, lane, xh, xl, aM, aC, [&](int kb) { epi_slice<false, ST>(pM, pC, kb, mw, dummy, yh[m - 1], yl[m - 1], ovf, (ST && st) ? st + F16_ST_M(m - 1) : nullptr, stscale); });
        pM = aM; pC = aC;
    }
    const uint32_t mw = (mk[3] >> 16) & 0xffffu;
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) epi_slice<false, ST>(pM, pC, kb, mw, dummy, yh[7], yl[7], ovf, (ST && st) ? st + F16_ST_M(7) : nullptr, stscale);
}

// MODE 0 (FULL): forward + reverse for every listed sample (stage API, train mode).
// MODE 1 (FWD) : forward only; the relu masks go to `masks` (224 B per sample, indexed by sample) and the samples
//                with sigma > 0 are appended to pos_list / pos_count.
// MODE 2 (BWD) : reverse pass only, for the samples of the given list, from the stored masks.
// The split exists for eval-mode rendering: a sample with sigma <= 0 has alpha = 1 - exp(-relu(sigma) dist) = 0
// exactly, its weight is 0 and its colour - hence d sigma/dx, the normal and the lighting MLP - is never used
// (utils/nerf_net_utils.py:18-39).  On the benchmark frame that is 71 % of the evaluated samples.
#define F16_FULL 0
#define F16_FWD 1
#define F16_BWD 2
#define F16_TRAIN 3           // FULL + every layer's activations and masked sigma-adjoints stored row-major for dsn_train.hip
#define F16_FIRST_BWD_BLOCK (OFF_L6T / DSN_BLK)   // 448
template <int MODE>
__global__ void __launch_bounds__(F16_THREADS, 1)
k_field16(const float* __restrict__ packed, const DsnFrameState* __restrict__ fs, const float* __restrict__ x_c,
          int64_t N, const int32_t* __restrict__ active_list, const int32_t* __restrict__ active_count,
          float* __restrict__ sigma, float* __restrict__ essence, float* __restrict__ grad,
          uint4* __restrict__ masks, int32_t* __restrict__ pos_list, int32_t* __restrict__ pos_count,
          float* __restrict__ tr_h, float* __restrict__ tr_a, float* __restrict__ tr_rr, int64_t slot_base, int64_t rec_cap,
          const int32_t* __restrict__ sel, int32_t* __restrict__ flag_count) {
    // flag_count (optional): incremented once per sample this launch flags for the exact-fp32 fallback (sigma = NaN); k_field<fix>
    //            looks at it first and leaves when nothing was flagged
    // slot_base: the launch works on list slots slot_base ... (FULL mode as the overflow pass of the eval split, see dsn_render_rays)
    // rec_cap  : capacity of the relu-record array in samples.  FWD / BWD index the records by the sample's slot on the
    //            sigma > 0 list (FWD writes a record only for the samples it appends there, BWD reads slot s of the list it
    //            walks); samples whose slot is >= rec_cap get no record and are left to the overflow pass.  TRAIN indexes by sample.
    constexpr bool ST = MODE == F16_TRAIN;
    // LDS map: weight ring 64 KB | relu masks 7 layers x 4 words x 256 threads = 28 KB | PE operands 8 x half8 x 256 = 32 KB
    __shared__ __attribute__((aligned(16))) char ring[F16_RING_SLOTS * 4096];
    __shared__ uint32_t s_mask[7][4][F16_THREADS];
    // every small vector the epilogues need (no compiler-visible global load may sit in the steady state: its
    // vmcnt wait would drain the LDS-DMA queue): [bias0 256 | OFF_B1.. 2304 (6 biases, rgb bias, W_den, W_rgb3) | scalars 8]
    __shared__ __attribute__((aligned(16))) float s_vec[256 + 2304 + 8];
    __shared__ __attribute__((aligned(16))) half8 s_pe[8][F16_THREADS];
    __shared__ int s_app[8];      // forward kernel: per-wave counts + base of the tile's append to the sigma > 0 list
#ifndef F16_SHARE_SIMD      // (experiments: -DF16_SHARE_SIMD restores rounds 1-4, where other kernels' waves could sit beside these)
    DSN_OWN_SIMD();
#endif
    const int tid0 = threadIdx.x;
    const int lane0 = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int64_t count = (active_list ? (int64_t)(*active_count) : N) - slot_base;
    if (MODE == F16_BWD && count > rec_cap) count = rec_cap;
    // persistent workgroups: tile t = list slots [128 t, 128 t + 128), dealt round-robin over the grid (one workgroup per CU).  The
    // small vectors are fetched once per workgroup; a launch sized for the worst case no longer pays for its empty workgroups
    const int64_t ntiles = (count + 127) / 128;
    if ((int64_t)blockIdx.x >= ntiles) return;   // block-uniform: the barriers below need all 4 waves
    for (int i = tid0; i < 256 + 2304 + 8; i += F16_THREADS)
        s_vec[i] = i < 256 ? fs->bias0[i] : (i < 2560 ? packed[OFF_B1 + (i - 256)] : packed[OFF_SCAL + (i - 2560)]);
    // accumulators of the forward pass start from 64 x bias (6 trunk biases + rgb_net.1 bias follow bias0)
    for (int i = tid0; i < 256 + (OFF_WDEN - OFF_B1); i += F16_THREADS) s_vec[i] *= F16_FWD_SCALE;
    const float* const v_bias0 = s_vec;
    const float* const v_b1 = s_vec + 256;                                  // + l * 256
    const float* const v_brgb1 = s_vec + 256 + (OFF_BRGB1 - OFF_B1);
    const float* const v_wden = s_vec + 256 + (OFF_WDEN - OFF_B1);
    const float* const v_wrgb3 = s_vec + 256 + (OFF_WRGB3 - OFF_B1);
    const float* const v_scal = s_vec + 2560;
    W16 w;
    w.g = reinterpret_cast<const char*>(packed + OFF16_BASE) + wave * 8192 + 4096 + lane0 * 16;
    w.ring = ring;
    w.ring_off = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)ring;
    w.wave = wave;
    if (MODE == F16_FWD) w.nchunk = F16_FIRST_BWD_BLOCK / F16_CHUNK;
    if (MODE == F16_TRAIN) w.spread = false;
    // this tile's point comes from the previous tile's prefetch: list slot -> (slot on the list, sample index), then its coordinates
    auto tile_point = [&](int64_t t, bool& ok, int64_t& ls) -> int64_t {
        int64_t sl = (t * 4 + wave) * 32 + (lane0 & 31);
        ok = sl < count;
        if (!ok) sl = count - 1;
        // sel (BWD, early-stop shading list): entry `sl` names the SLOT of the sample on active_list (= the sigma > 0 list), which
        // is also where its relu record lies; *active_count is then the length of sel
        ls = sel ? (int64_t)sel[sl] : slot_base + sl;
        return active_list ? (int64_t)active_list[ls] : ls;
    };
    bool valid_n;
    int64_t lslot_n;
    int64_t pt_n = tile_point(blockIdx.x, valid_n, lslot_n);
    float xn[3] = {x_c[3 * pt_n], x_c[3 * pt_n + 1], x_c[3 * pt_n + 2]};
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    // (opaque copies: the unrolled body holds hundreds of LDS addresses derived from the thread index and one DMA source address per
    //  chunk derived from w.g; loop-invariant, the compiler would hoist them all out of the tile loop and spill)
    int tid = tid0, lane = lane0;
    asm volatile("" : "+v"(tid), "+v"(lane), "+v"(w.g));
    const int half = lane >> 5;
#if F16_TIMING
    unsigned long long tstamp[6], rstamp[6];
#endif
    F16_STAMP(0);
    w16_begin_issue(w, MODE == F16_BWD ? F16_FIRST_BWD_BLOCK : 0);   // first: the weights' trip from L2 runs under the rest of the prologue
#if !F16_NEXT_POINT
    if (tile != (int64_t)blockIdx.x) { pt_n = tile_point(tile, valid_n, lslot_n); xn[0] = x_c[3 * pt_n]; xn[1] = x_c[3 * pt_n + 1]; xn[2] = x_c[3 * pt_n + 2]; }
#endif
    const bool valid = valid_n;
    const int64_t lslot = lslot_n, pt = pt_n;
    const float xa[3] = {xn[0], xn[1], xn[2]};
    const bool more = F16_NEXT_POINT && MODE != F16_TRAIN && tile + gridDim.x < ntiles;      // workgroup-uniform
    if (more) pt_n = tile_point(tile + gridDim.x, valid_n, lslot_n);       // (its coordinates follow one layer into the tile: NEXT_POINT)
#define NEXT_POINT() do { if (more) { xn[0] = x_c[3 * pt_n]; xn[1] = x_c[3 * pt_n + 1]; xn[2] = x_c[3 * pt_n + 2]; } } while (0)
    w16_begin_wait(w, lane, MODE == F16_BWD ? F16_FIRST_BWD_BLOCK : 0);
    int blk = MODE == F16_BWD ? F16_FIRST_BWD_BLOCK : 0;
    F16_STAMP(1);

    // relu masks live in LDS between the forward and the reverse pass (28 VGPRs otherwise); lane-private slots,
    // so no barrier is needed around them
#define MK_STORE(L, mk) { s_mask[L][0][tid] = mk[0]; s_mask[L][1][tid] = mk[1]; s_mask[L][2][tid] = mk[2]; s_mask[L][3][tid] = mk[3]; }
#define MK_LOAD(L, mk) { mk[0] = s_mask[L][0][tid]; mk[1] = s_mask[L][1][tid]; mk[2] = s_mask[L][2][tid]; mk[3] = s_mask[L][3][tid]; }
    uint32_t mk[4];
    half8 ah[8][2], al[8][2], bh[8][2], bl[8][2];
    float ovf = 0.0f;          // range guard: running max of |value| over everything this lane splits into fp16
    // per-sample mask record: [half][layer] uint4, 224 B contiguous per sample
    // TRAIN: indexed by sample; BWD: by the slot of the list it walks; FWD: by the slot the sample gets on the sigma > 0 list (below)
    uint4* mrec = masks ? masks + ((size_t)(MODE == F16_BWD ? lslot : pt) * 2 + half) * 7 : nullptr;
    // training kernel: this lane's row in the row-major [N,256] activation arrays (layer stride N * 256 floats)
    const int64_t tr_ls = N * 256;
#if defined(DSN_EXPERIMENTS) && defined(F16_TRAIN_ABL)      // timing ablation (WRONG results): 1 = no h_l stores, 2 = no a_l stores,
    // 8 / 16: the same store instructions into a SMALL footprint (rows folded onto 4096 / 256 rows per layer: 57 MB - inside the 256 MB
    // Infinity Cache - / 3.6 MB - inside one L2): is the cost of the stores the write-back to HBM or their issue?
    const int64_t pt_st = (F16_TRAIN_ABL & 8) ? (pt & 4095) : ((F16_TRAIN_ABL & 16) ? (pt & 255) : pt);
    const int64_t sl_st = (tile * 4 + wave) * 32 + (lane & 31);      // (& 32: tile-major emulation, by list slot)
    const int64_t row_off = (F16_TRAIN_ABL & 32) ? (sl_st >> 5) * 8192 + (sl_st & 31) * 32 : pt_st * 256;
    float* const th = ST && valid && !(F16_TRAIN_ABL & 1) ? tr_h + row_off + 4 * half : nullptr;
    float* const ta = ST && valid && !(F16_TRAIN_ABL & 2) ? tr_a + row_off + 4 * half : nullptr;
#else
    float* const th = ST && valid ? tr_h + pt * 256 + 4 * half : nullptr;     // + l * tr_ls : h_l
    float* const ta = ST && valid ? tr_a + pt * 256 + 4 * half : nullptr;     // + l * tr_ls : masked sigma-adjoint of layer l
#endif

  if (MODE != F16_BWD) {
    // positional encoding (fp32, accurate sincos) -> split k-steps; same slot map as k_field
    half8 ph[2][2], pl[2][2];
    {
        f32x16 pe[2];
#pragma unroll
        for (int t = 0; t < 30; ++t) {
            float s, c;
#if F16_SINCOS_OCML
            sincosf(xa[t % 3] * (float)(1 << (t / 3)), &s, &c);
#else
            dsn_sincos(xa[t % 3] * (float)(1 << (t / 3)), s, c);
#endif
            pe[t >> 4][t & 15] = half ? c : s;
        }
        pe[1][14] = half ? xa[1] : xa[0];
        pe[1][15] = half ? 0.0f : xa[2];
        split16<false>(pe[0], ph[0], pl[0]);
        split16<false>(pe[1], ph[1], pl[1]);
        s_pe[0][tid] = ph[0][0]; s_pe[1][tid] = ph[0][1]; s_pe[2][tid] = ph[1][0]; s_pe[3][tid] = ph[1][1];
        s_pe[4][tid] = pl[0][0]; s_pe[5][tid] = pl[0][1]; s_pe[6][tid] = pl[1][0]; s_pe[7][tid] = pl[1][1];
    }

    F16_STAMP(2);
    // stage1.0
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        f32x16 aM = rows16(v_bias0, m, half), aC = zero16();
        dense16<2, false>(w, blk, lane, ph, pl, aM, aC);
        f32x16 v = unscale16(aM, aC);
        const uint32_t bits = relu_bits16(v);
        if (m & 1) mk[m >> 1] |= bits << 16; else mk[m >> 1] = bits;
        if (ST) store16(th ? th + 0 * tr_ls + F16_ST_M(m) : nullptr, v, 1.0f);
        track16(ovf, v);
        split16<false>(v, ah[m], al[m]);
    }
    MK_STORE(0, mk)
    layer16_fwd<ST>(w, blk, lane, v_b1 + 0 * 256, ah, al, bh, bl, mk, ovf, th ? th + 1 * tr_ls : nullptr); MK_STORE(1, mk)
    NEXT_POINT();
    layer16_fwd<ST>(w, blk, lane, v_b1 + 1 * 256, bh, bl, ah, al, mk, ovf, th ? th + 2 * tr_ls : nullptr); MK_STORE(2, mk)
    layer16_fwd<ST>(w, blk, lane, v_b1 + 2 * 256, ah, al, bh, bl, mk, ovf, th ? th + 3 * tr_ls : nullptr); MK_STORE(3, mk)
#if F16_PIPE_LATE
    // stage2.0 : [h, pe] -> 256
    layer16_fwd<ST, true, false>(w, blk, lane, v_b1 + 3 * 256, bh, bl, ah, al, mk, ovf, th ? th + 4 * tr_ls : nullptr, s_pe);
    MK_STORE(4, mk)
    layer16_fwd<ST>(w, blk, lane, v_b1 + 4 * 256, ah, al, bh, bl, mk, ovf, th ? th + 5 * tr_ls : nullptr); MK_STORE(5, mk)
    // stage2.4 with the density head fused into its epilogue
    float sg_part = 0.0f;
    layer16_fwd<ST, false, true>(w, blk, lane, v_b1 + 5 * 256, bh, bl, ah, al, mk, ovf, th ? th + 6 * tr_ls : nullptr, nullptr, v_wden,
                                 &sg_part);
#else
    // (round 1's form, kept for A/B runs: the epilogues of stage2.0 and stage2.4 behind their MFMAs, not under the next block's)
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        f32x16 aM = rows16(v_b1 + 3 * 256, m, half), aC = zero16();
        dense16<8, false>(w, blk, lane, bh, bl, aM, aC);
        {
            half8 qh[2][2], ql[2][2];
            qh[0][0] = s_pe[0][tid]; qh[0][1] = s_pe[1][tid]; qh[1][0] = s_pe[2][tid]; qh[1][1] = s_pe[3][tid];
            ql[0][0] = s_pe[4][tid]; ql[0][1] = s_pe[5][tid]; ql[1][0] = s_pe[6][tid]; ql[1][1] = s_pe[7][tid];
            dense16<2, false>(w, blk, lane, qh, ql, aM, aC);
        }
        f32x16 v = unscale16(aM, aC);
        const uint32_t bits = relu_bits16(v);
        if (m & 1) mk[m >> 1] |= bits << 16; else mk[m >> 1] = bits;
        if (ST) store16(th ? th + 4 * tr_ls + F16_ST_M(m) : nullptr, v, 1.0f);
        track16(ovf, v);
        split16<false>(v, ah[m], al[m]);
    }
    MK_STORE(4, mk)
    layer16_fwd<ST>(w, blk, lane, v_b1 + 4 * 256, ah, al, bh, bl, mk, ovf, th ? th + 5 * tr_ls : nullptr); MK_STORE(5, mk)
    float sg_part = 0.0f;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        f32x16 aM = rows16(v_b1 + 5 * 256, m, half), aC = zero16();
        dense16<8, false>(w, blk, lane, bh, bl, aM, aC);
        f32x16 v = unscale16(aM, aC);
        const uint32_t bits = relu_bits16(v);
        if (m & 1) mk[m >> 1] |= bits << 16; else mk[m >> 1] = bits;
        if (ST) store16(th ? th + 6 * tr_ls + F16_ST_M(m) : nullptr, v, 1.0f);
        const f32x16 wd = rows16(v_wden, m, half);
#pragma unroll
        for (int r = 0; r < 16; ++r) sg_part = fmaf(wd[r], v[r], sg_part);
        track16(ovf, v);
        split16<false>(v, ah[m], al[m]);
    }
#endif
    F16_STAMP(3);
    sg_part += __shfl_xor(sg_part, 32);
    const float sg = sg_part + v_scal[0];
    // range guard, forward half: a flagged sample carries sigma = NaN until the exact-fp32 kernel has re-evaluated it
    const bool flag_fwd = !(fmaxf(ovf, __shfl_xor(ovf, 32)) < F16_RANGE);
    if (valid && half == 0) sigma[pt] = flag_fwd ? dsn_nan_flag() : sg;
    if (MODE != F16_TRAIN && flag_fwd && valid && half == 0 && flag_count) atomicAdd(flag_count, 1);
    bool write_rec = ST && valid;
    if (MODE == F16_FWD) {
        // samples with positive density -> the reverse-pass list (flagged samples too: the fallback behind the reverse pass
        // walks this list, and whatever it finds their density to be, a normal and a colour for them cost nothing but time)
        const bool pos = valid && half == 0 && (sg > 0.0f || flag_fwd);
        const unsigned long long bm = __ballot(pos);
        const int cnt = __popcll(bm);
        // workgroup-aggregated append: the tile's samples stay together on the sigma > 0 list (they are neighbours in the cell-major
        // order of the nearest-face sort, and k_normal / the reverse pass / k_light16 read their lists 64 - 128 entries per wave /
        // workgroup: per-wave appends interleave 32-sample pieces of different tiles)
        if (lane == 0) s_app[wave] = cnt;
        __syncthreads();
        if (tid == 0) {
            const int tot = s_app[0] + s_app[1] + s_app[2] + s_app[3];
            s_app[4] = tot ? atomicAdd(pos_count, tot) : 0;
        }
        __syncthreads();
        int base = s_app[4];
        for (int k = 0; k < wave; ++k) base += s_app[k];
        const int my = base + __popcll(bm & ((1ull << lane) - 1ull));
        if (pos) pos_list[my] = (int32_t)pt;
        // its relu record goes to the same slot (both half-waves hold a half of it); none beyond the capacity
        const int rslot = __shfl(my, lane & 31);
        write_rec = ((bm >> (lane & 31)) & 1ull) != 0 && (int64_t)rslot < rec_cap;
        mrec = masks + ((size_t)rslot * 2 + half) * 7;
    }
    if (MODE == F16_FWD || MODE == F16_TRAIN) {
        // masks of all 7 layers -> the sample's record (read back by k_field16<reverse> / k_tangent16)
        MK_STORE(6, mk)
        if (write_rec) {
#pragma unroll
            for (int L = 0; L < 7; ++L)
                mrec[L] = make_uint4(s_mask[L][0][tid], s_mask[L][1][tid], s_mask[L][2][tid], s_mask[L][3][tid]);
        }
    }
    // rgb_net: 256 -> 128 -> relu -> 3 (second layer as per-lane dots in the epilogue)
    {
        float e0 = 0.f, e1 = 0.f, e2 = 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            f32x16 aM = rows16(v_brgb1, m, half), aC = zero16();
            dense16<8, false>(w, blk, lane, ah, al, aM, aC);
            const f32x16 v = unscale16(aM, aC);
            const f32x16 w0 = rows16(v_wrgb3 + 0 * 128, m, half);
            const f32x16 w1 = rows16(v_wrgb3 + 1 * 128, m, half);
            const f32x16 w2 = rows16(v_wrgb3 + 2 * 128, m, half);
            if (ST && valid) {      // rgb_net.1 output after its relu, row-major [N,128]
                f32x16 xr;
#pragma unroll
                for (int r = 0; r < 16; ++r) xr[r] = fmaxf(v[r], 0.0f);
                store16(tr_rr + pt * 128 + 4 * half + 32 * m, xr, 1.0f);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float x = fmaxf(v[r], 0.0f);
                e0 = fmaf(w0[r], x, e0); e1 = fmaf(w1[r], x, e1); e2 = fmaf(w2[r], x, e2);
            }
        }
        e0 += __shfl_xor(e0, 32); e1 += __shfl_xor(e1, 32); e2 += __shfl_xor(e2, 32);
        if (valid && half == 0) {
            essence[3 * pt + 0] = e0 + v_scal[1];
            essence[3 * pt + 1] = e1 + v_scal[2];
            essence[3 * pt + 2] = e2 + v_scal[3];
        }
    }

#if F16_TIMING
    F16_STAMP(4);
    if (MODE == F16_FWD && tid == 0) {
        for (int i = 0; i < 4; ++i) { atomicAdd(&g_f16_timing[i], tstamp[i + 1] - tstamp[i]); atomicAdd(&g_f16_timing[8 + i], rstamp[i + 1] - rstamp[i]); }
        atomicAdd(&g_f16_timing[7], 1ull);
    }
#endif
    if (MODE == F16_FWD) { __syncthreads(); continue; }      // (tile done: every wave has read its last weight block)
  } else {
    // MODE == BWD: masks come back from the sample's record
#pragma unroll
    for (int L = 0; L < 7; ++L) {
        const uint4 q = mrec[L];
        s_mask[L][0][tid] = q.x; s_mask[L][1][tid] = q.y; s_mask[L][2][tid] = q.z; s_mask[L][3][tid] = q.w;
    }
    MK_LOAD(6, mk)
  }

    // ---- reverse pass on g / 64: seed = W_den masked by relu(stage2.4)
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        f32x16 g = rows16(v_wden, m, half);
#pragma unroll
        for (int r = 0; r < 16; ++r) g[r] *= F16_GSCALE;
        mask16(g, (mk[m >> 1] >> (16 * (m & 1))) & 0xffffu);
        if (ST) store16(ta ? ta + 6 * tr_ls + F16_ST_M(m) : nullptr, g, F16_GUNSCALE);
        split16<true>(g, ah[m], al[m]);
    }
    MK_LOAD(5, mk) layer16_bwd<ST>(w, blk, lane, ah, al, bh, bl, mk, ovf, ta ? ta + 5 * tr_ls : nullptr);
    if (MODE == F16_BWD) NEXT_POINT();
    MK_LOAD(4, mk) layer16_bwd<ST>(w, blk, lane, bh, bl, ah, al, mk, ovf, ta ? ta + 4 * tr_ls : nullptr);
    MK_LOAD(3, mk)
    // stage2.0^T : 256 -> [256 h | 64 pe]  (epilogues not pipelined: as a layer16_bwd the reverse-only kernel spills 267 registers)
    f32x16 dpe[2];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        f32x16 aM = zero16(), aC = zero16();
        dense16<8, true>(w, blk, lane, ah, al, aM, aC);
        f32x16 v = fold16(aM, aC);
        mask16(v, (mk[m >> 1] >> (16 * (m & 1))) & 0xffffu);
        if (ST) store16(ta ? ta + 3 * tr_ls + F16_ST_M(m) : nullptr, v, F16_GUNSCALE);
        track16(ovf, v);
        split16<true>(v, bh[m], bl[m]);
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        f32x16 aM = zero16(), aC = zero16();
        dense16<8, true>(w, blk, lane, ah, al, aM, aC);
        dpe[b] = fold16(aM, aC);
    }
    MK_LOAD(2, mk) layer16_bwd<ST>(w, blk, lane, bh, bl, ah, al, mk, ovf, ta ? ta + 2 * tr_ls : nullptr);
    MK_LOAD(1, mk) layer16_bwd<ST>(w, blk, lane, ah, al, bh, bl, mk, ovf, ta ? ta + 1 * tr_ls : nullptr);
    MK_LOAD(0, mk) layer16_bwd<ST>(w, blk, lane, bh, bl, ah, al, mk, ovf, ta ? ta + 0 * tr_ls : nullptr);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        f32x16 aM = zero16(), aC = zero16();
        dense16<8, true>(w, blk, lane, ah, al, aM, aC);
        const f32x16 v = fold16(aM, aC);
#pragma unroll
        for (int r = 0; r < 16; ++r) dpe[b][r] += v[r];
    }

    // encoding backward (each lane re-derives its own sin / cos: both are needed for the derivative)
    {
        // (opaque copy of the point: otherwise the forward pass's 60 sines and cosines are kept alive - spilled - across the tile)
        float xb[3] = {xa[0], xa[1], xa[2]};
        asm volatile("" : "+v"(xb[0]), "+v"(xb[1]), "+v"(xb[2]));
        float g[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 30; ++t) {
            const int j = t / 3, a = t % 3;
            float s, c;
#if F16_SINCOS_OCML
            sincosf(xb[a] * (float)(1 << j), &s, &c);
#else
            dsn_sincos(xb[a] * (float)(1 << j), s, c);
#endif
            const float d = dpe[t >> 4][t & 15];
            const float term = (d * (half ? s : c)) * (float)(1 << j);
            g[a] += half ? -term : term;
        }
        const float i30 = dpe[1][14], i31 = dpe[1][15];
        if (half) g[1] += i30; else { g[0] += i30; g[2] += i31; }
        g[0] += __shfl_xor(g[0], 32); g[1] += __shfl_xor(g[1], 32); g[2] += __shfl_xor(g[2], 32);
        if (valid && half == 0) {
            grad[3 * pt] = g[0] * F16_GUNSCALE; grad[3 * pt + 1] = g[1] * F16_GUNSCALE; grad[3 * pt + 2] = g[2] * F16_GUNSCALE;
        }
    }
    // range guard, reverse half (ovf still holds the forward maximum in the single-launch modes)
    const bool flagged = !(fmaxf(ovf, __shfl_xor(ovf, 32)) < F16_RANGE);
    if (flagged && valid && half == 0) {
        if (MODE == F16_TRAIN) { if (pos_count) atomicAdd(pos_count, 1); }    // training: counted, reported by the host mirror
        else { sigma[pt] = dsn_nan_flag(); if (flag_count) atomicAdd(flag_count, 1); }
    }
    if (MODE == F16_TRAIN) break;   // (the training kernel keeps one tile per workgroup: its extra live state leaves no room for the loop's)
    __syncthreads();      // tile done: the ring is free for the next tile's first chunk
  }
}

void dsn_launch_field16(const float* packed, const DsnFrameState* fs, const float* x_c, int64_t N,
                        const int32_t* active_list, const int32_t* active_count, float* sigma, float* essence,
                        float* grad, hipStream_t st, int32_t* flag_count) {
    int64_t blocks = (N + 127) / 128;
    if (blocks == 0) return;
    hipLaunchKernelGGL(k_field16<F16_FULL>, dim3((unsigned)std::min<int64_t>(blocks, dsn_cu_count())), dim3(F16_THREADS), 0, st, packed, fs, x_c, N,
                       active_list, active_count, sigma, essence, grad, (uint4*)nullptr, (int32_t*)nullptr,
                       (int32_t*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, (int64_t)0, (int64_t)0, (const int32_t*)nullptr,
                       flag_count);
}
// the same single-launch evaluation on slots slot_base ... of a list: the overflow pass of the eval split (samples of the
// sigma > 0 list whose relu record did not fit get forward AND reverse here; identical values, see dsn_render_rays)
void dsn_launch_field16_from(const float* packed, const DsnFrameState* fs, const float* x_c, int64_t N, const int32_t* list,
                             const int32_t* count, int64_t slot_base, float* sigma, float* essence, float* grad, hipStream_t st,
                             int32_t* flag_count) {
    int64_t blocks = (N - slot_base + 127) / 128;
    if (blocks <= 0) return;
    hipLaunchKernelGGL(k_field16<F16_FULL>, dim3((unsigned)std::min<int64_t>(blocks, dsn_cu_count())), dim3(F16_THREADS), 0, st, packed, fs, x_c, N, list, count, sigma,
                       essence, grad, (uint4*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, (float*)nullptr, (float*)nullptr,
                       (float*)nullptr, slot_base, (int64_t)0, (const int32_t*)nullptr, flag_count);
}
// training: dense evaluation that also leaves h_l [7][N,256], the masked sigma-adjoints a_l [7][N,256] and the rgb hidden
// layer [N,128] in row-major arrays for the weight-gradient products of dsn_train.hip
void dsn_launch_field16_train(const float* packed, const DsnFrameState* fs, const float* x_c, int64_t N, float* sigma,
                              float* essence, float* grad, float* tr_h, float* tr_a, float* tr_rr, void* masks, hipStream_t st,
                              int32_t* range_count, const int32_t* row_list, const int32_t* row_count) {
    int64_t blocks = (N + 127) / 128;
    if (blocks == 0) return;
    // range_count (optional): incremented once per sample whose activations / adjoints left the fp16 range
    // row_list / row_count (optional): only the listed samples are evaluated (their rows are written, the others left alone)
    hipLaunchKernelGGL(k_field16<F16_TRAIN>, dim3((unsigned)blocks), dim3(F16_THREADS), 0, st, packed, fs, x_c, N,
                       row_list, row_count, sigma, essence, grad, (uint4*)masks, (int32_t*)nullptr,
                       range_count, tr_h, tr_a, tr_rr, (int64_t)0, N, (const int32_t*)nullptr, (int32_t*)nullptr);
}
// eval-mode split: forward on the active samples (+ masks, + list of sigma > 0 samples) ...
void dsn_launch_field16_fwd(const float* packed, const DsnFrameState* fs, const float* x_c, int64_t N,
                            const int32_t* active_list, const int32_t* active_count, float* sigma, float* essence,
                            void* masks, int32_t* pos_list, int32_t* pos_count, hipStream_t st, int64_t rec_cap, int32_t* flag_count) {
    int64_t blocks = (N + 127) / 128;
    if (blocks == 0) return;
    hipLaunchKernelGGL(k_field16<F16_FWD>, dim3((unsigned)std::min<int64_t>(blocks, dsn_cu_count())), dim3(F16_THREADS), 0, st, packed, fs, x_c, N,
                       active_list, active_count, sigma, essence, (float*)nullptr, (uint4*)masks, pos_list, pos_count,
                       (float*)nullptr, (float*)nullptr, (float*)nullptr, (int64_t)0, rec_cap, (const int32_t*)nullptr, flag_count);
}
// ... reverse pass on the sigma > 0 samples only
void dsn_launch_field16_bwd(const float* packed, const DsnFrameState* fs, const float* x_c, int64_t N,
                            const int32_t* pos_list, const int32_t* pos_count, float* grad, const void* masks,
                            hipStream_t st, float* sigma, int64_t rec_cap, const int32_t* sel, const int32_t* sel_count,
                            int32_t* flag_count) {
    // sel / sel_count (DSN_EARLY_STOP): walk only the listed slots of pos_list (all below rec_cap); N bounds their number
    int64_t blocks = ((rec_cap < N ? rec_cap : N) + 127) / 128;
    if (blocks == 0) return;
    // sigma: only ever WRITTEN here, with the NaN flag of a sample whose adjoints left the fp16 range (see the header)
    hipLaunchKernelGGL(k_field16<F16_BWD>, dim3((unsigned)std::min<int64_t>(blocks, dsn_cu_count())), dim3(F16_THREADS), 0, st, packed, fs, x_c, N,
                       pos_list, sel ? sel_count : pos_count, sigma, (float*)nullptr, grad, (uint4*)masks, (int32_t*)nullptr,
                       (int32_t*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, (int64_t)0, rec_cap, sel, flag_count);
}

// ---------------------------------------------------------------------------------------------
// k_tangent16 : forward TANGENT pass of the trunk for the training backward (dsn_train.hip): hdot_l = m_l * (W_l hdot_{l-1}),
// hdot_0 = m_0 * (W_0[:,pe] J_pe u) with u = dL/d(d sigma/dx) per sample and m_l the relu patterns the training forward
// recorded.  The network is linear in u, so every sample is evaluated on u / max|u| (O(1) operands for the split-fp16
// products, forward flavour of dense16) and its outputs are stored times max|u|: row-major hdot_l [7][N,256].
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void epi_slice_tan(const f32x16& pM, const f32x16& pC, int kb, uint32_t mword, half8 (&yh)[2],
                                              half8 (&yl)[2], float* st, float stscale, float& ovf, float* keep = nullptr) {
    float vv[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int r = 2 * kb + e;
        const float v = dsn_keep_active((pM[r] + pC[r]) * F16_FWD_INV, mword, r);
        vv[e] = v;
        const _Float16 hi = (_Float16)v;
        yh[r >> 3][r & 7] = hi;
        yl[r >> 3][r & 7] = (_Float16)fmaf((float)hi, -1.0f, v);
    }
    asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(ovf) : "v"(vv[0]), "v"(vv[1]));      // range guard (see k_tangent16)
#if defined(DSN_EXPERIMENTS) && defined(F16_TRAIN_ABL) && (F16_TRAIN_ABL & 256)
    if (kb == 7) {
        float o[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = dsn_keep_active((pM[r] + pC[r]) * F16_FWD_INV, mword, r) * stscale;
        store16_lines_emul(st, o);
        return;
    }
#endif
    if (keep && kb == 7) {    // (round 6, the last layer: the block's 16 values stay in registers for the column sums - k_tangent16)
#pragma unroll
        for (int r = 0; r < 16; ++r) keep[r] = dsn_keep_active((pM[r] + pC[r]) * F16_FWD_INV, mword, r) * stscale;
        return;
    }
    if (st && kb == 7) {      // the block's 16 values again, stored together as four 16-byte pieces (see epi_slice; 8-byte pieces slice by
                              // slice: 1.18 -> 1.09 ms)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = dsn_keep_active((pM[4 * q + e] + pC[4 * q + e]) * F16_FWD_INV, mword, 4 * q + e) * stscale;
            *reinterpret_cast<float4*>(st + 8 * q) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}
__device__ __forceinline__ void layer16_tan(W16& w, int& blk, int lane, const half8 (&xh)[8][2], const half8 (&xl)[8][2],
                                            half8 (&yh)[8][2], half8 (&yl)[8][2], const uint32_t (&mk)[4], float* st, float stscale,
                                            float& ovf, float (*keep)[16] = nullptr) {
    // keep (the last layer, round 6): the outputs stay in registers, [block][accumulator register], instead of being stored
    f32x16 pM = zero16(), pC = zero16();
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        f32x16 aM = zero16(), aC = zero16();
        const uint32_t mw = m > 0 ? ((mk[(m - 1) >> 1] >> (16 * ((m - 1) & 1))) & 0xffffu) : 0u;
        if (m == 0) dense16<8, false>(w, blk, lane, xh, xl, aM, aC);
        else dense16<8, false>(w, blk, lane, xh, xl, aM, aC, [&](int kb) { epi_slice_tan(pM, pC, kb, mw, yh[m - 1], yl[m - 1], st ? st + 32 * (m - 1) : nullptr, stscale, ovf, keep ? keep[m - 1] : nullptr); });
        pM = aM; pC = aC;
    }
    const uint32_t mw = (mk[3] >> 16) & 0xffffu;
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) epi_slice_tan(pM, pC, kb, mw, yh[7], yl[7], st ? st + 32 * 7 : nullptr, stscale, ovf, keep ? keep[7] : nullptr);
}

// sum of v over the 32 lanes of a half-wave, left in its last lane (31 / 63): four row-shift adds inside the rows of 16 + one row
// broadcast (lane 15 of rows 0 / 2 into rows 1 / 3) - five VALU instructions with DPP operands, no LDS crossbar
__device__ __forceinline__ float half_wave_sum_to_last(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));      // row_shr:1
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));      // row_shr:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));      // row_shr:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));      // row_shr:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, true));      // row_bcast:15 into rows 1, 3
    return v;
}

// A sample whose tangent / adjoint left the fp16 range (range guard of k_tangent16 / k_adjoint16): its rows of all `layers` output
// arrays are rewritten as zeros - the sample drops out of this step's second-order / adjoint weight gradients instead of putting
// inf / NaN into them - and it is counted (the host mirror warns).  `t` = this lane's slot of layer 0, `ls` = layer stride.
__device__ __forceinline__ void zero_train_rows(float* t, int64_t ls, int layers) {
    for (int L = 0; L < layers; ++L)
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(t + L * ls + 32 * m + 8 * q) = make_float4(0.f, 0.f, 0.f, 0.f);
}

__global__ void __launch_bounds__(F16_THREADS, 1)
k_tangent16(const float* __restrict__ packed, const float* __restrict__ x_c, const float* __restrict__ u, int64_t N,
            const uint4* __restrict__ masks, float* __restrict__ tr_t, uint32_t* __restrict__ gmax, int32_t* __restrict__ range_count,
            const int32_t* __restrict__ row_list, const int32_t* __restrict__ row_count, float* __restrict__ colsum6) {
    // colsum6 (optional, round 6) [256]: += the column sums of hdot_6 over the listed samples - the only use the training backward has
    // for the last layer's tangent (d (w_d . hdot_6) / d w_d, dsn_train.hip).  With it the layer is not stored at all: one 1 KB row per
    // sample less to write and to read back, and no k_t_colsum launch (0.75 GB and 0.13 ms per 8192 x 64 step).
    // row_list / row_count (optional): the pass runs on the listed samples only (rows whose cotangents are all zero add nothing to
    // any gradient: dsn_train.hip, Rows); arrays stay indexed by sample, N is still the layer stride
    __shared__ __attribute__((aligned(16))) char ring[F16_RING_SLOTS * 4096];
    __shared__ uint32_t s_mask[7][4][F16_THREADS];
    __shared__ __attribute__((aligned(16))) half8 s_pe[8][F16_THREADS];
    DSN_OWN_SIMD_T(1);
    const int tid = threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5;
    const int64_t NL = row_count ? (int64_t)(*row_count) : N;
    if ((int64_t)blockIdx.x * 128 >= NL) return;      // block-uniform
    int64_t pt = ((int64_t)blockIdx.x * 4 + wave) * 32 + (lane & 31);
    const bool valid = pt < NL;
    if (!valid) pt = NL - 1;
    if (row_list) pt = (int64_t)row_list[pt];
    const float xa[3] = {x_c[3 * pt], x_c[3 * pt + 1], x_c[3 * pt + 2]};
    float ua[3] = {u[3 * pt], u[3 * pt + 1], u[3 * pt + 2]};
    // Per-sample scale.  The sample runs on u * 2^-6 / max|u|: the 2^-6 leaves the tangent of every hidden unit 64 x 65 000 of
    // head-room per unit of u before an fp16 operand overflows (the PE tangent alone is up to 512 x).  A max|u| below 1e-30 - a
    // cotangent that has underflowed to a denormal, as happens on samples whose weight is ~0 - counts as zero: 1 / max|u| would
    // overflow to inf and fill the weight gradients with NaN (found by training w4; round 2 divided unconditionally).
    float sc = fmaxf(fmaxf(fabsf(ua[0]), fabsf(ua[1])), fabsf(ua[2]));
    if (!(sc > 1e-30f) || !(sc < 3.0e38f)) sc = 0.0f;
    const float inv = sc > 0.0f ? 0.015625f / sc : 0.0f;
    for (int c = 0; c < 3; ++c) ua[c] = sc > 0.0f ? ua[c] * inv : 0.0f;
    sc *= 64.0f;                                  // what the stored outputs are multiplied back by
    float ovf = 0.0f;                             // range guard: running max of |value| over everything this lane splits into fp16
    if (gmax) {      // batch-wide magnitude of the outputs (bit pattern of a non-negative float orders like the float)
        float wm = valid ? sc * 0.015625f : 0.0f;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) wm = fmaxf(wm, __shfl_xor(wm, o));
        if (lane == 0 && __float_as_uint(wm) > __atomic_load_n(gmax, __ATOMIC_RELAXED)) atomicMax(gmax, __float_as_uint(wm));
    }
    {
        const uint4* mrec = masks + ((size_t)pt * 2 + half) * 7;
#pragma unroll
        for (int L = 0; L < 7; ++L) {
            const uint4 q = mrec[L];
            s_mask[L][0][tid] = q.x; s_mask[L][1][tid] = q.y; s_mask[L][2][tid] = q.z; s_mask[L][3][tid] = q.w;
        }
    }
    W16 w;
    w.g = reinterpret_cast<const char*>(packed + OFF16_BASE) + wave * 8192 + 4096 + lane * 16;
    w.ring = ring;
    w.ring_off = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)ring;
    w.wave = wave;
    w.spread = false;
    w16_begin(w, lane, 0);
    int blk = 0;
    const int64_t ls = N * 256;
    float* const tt = valid ? tr_t + pt * 256 + 4 * half : nullptr;
#define TMK_LOAD(L, mk) { mk[0] = s_mask[L][0][tid]; mk[1] = s_mask[L][1][tid]; mk[2] = s_mask[L][2][tid]; mk[3] = s_mask[L][3][tid]; }
    uint32_t mk[4];
    half8 ah[8][2], al[8][2], bh[8][2], bl[8][2];
    half8 ph[2][2], pl[2][2];
    {   // tangent of the encoding in the forward slot map: slot t holds (half ? cos : sin)(2^j x_a) -> (half ? -sin : cos) 2^j u_a
        f32x16 pe[2];
#pragma unroll
        for (int t = 0; t < 30; ++t) {
            float s, c;
            const float f = (float)(1 << (t / 3));
            dsn_sincos(xa[t % 3] * f, s, c);
            pe[t >> 4][t & 15] = (half ? -s : c) * f * ua[t % 3];
        }
        pe[1][14] = half ? ua[1] : ua[0];
        pe[1][15] = half ? 0.0f : ua[2];
        split16<false>(pe[0], ph[0], pl[0]);
        split16<false>(pe[1], ph[1], pl[1]);
        s_pe[0][tid] = ph[0][0]; s_pe[1][tid] = ph[0][1]; s_pe[2][tid] = ph[1][0]; s_pe[3][tid] = ph[1][1];
        s_pe[4][tid] = pl[0][0]; s_pe[5][tid] = pl[0][1]; s_pe[6][tid] = pl[1][0]; s_pe[7][tid] = pl[1][1];
    }
    TMK_LOAD(0, mk)
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        f32x16 aM = zero16(), aC = zero16();
        dense16<2, false>(w, blk, lane, ph, pl, aM, aC);
        f32x16 v = unscale16(aM, aC);
        mask16(v, (mk[m >> 1] >> (16 * (m & 1))) & 0xffffu);
        track16(ovf, v);
        store16(tt ? tt + 0 * ls + 32 * m : nullptr, v, sc);
        split16<false>(v, ah[m], al[m]);
    }
    TMK_LOAD(1, mk) layer16_tan(w, blk, lane, ah, al, bh, bl, mk, tt ? tt + 1 * ls : nullptr, sc, ovf);
    TMK_LOAD(2, mk) layer16_tan(w, blk, lane, bh, bl, ah, al, mk, tt ? tt + 2 * ls : nullptr, sc, ovf);
    TMK_LOAD(3, mk) layer16_tan(w, blk, lane, ah, al, bh, bl, mk, tt ? tt + 3 * ls : nullptr, sc, ovf);
    TMK_LOAD(4, mk)
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        f32x16 aM = zero16(), aC = zero16();
        dense16<8, false>(w, blk, lane, bh, bl, aM, aC);
        {
            half8 qh[2][2], ql[2][2];
            qh[0][0] = s_pe[0][tid]; qh[0][1] = s_pe[1][tid]; qh[1][0] = s_pe[2][tid]; qh[1][1] = s_pe[3][tid];
            ql[0][0] = s_pe[4][tid]; ql[0][1] = s_pe[5][tid]; ql[1][0] = s_pe[6][tid]; ql[1][1] = s_pe[7][tid];
            dense16<2, false>(w, blk, lane, qh, ql, aM, aC);
        }
        f32x16 v = unscale16(aM, aC);
        mask16(v, (mk[m >> 1] >> (16 * (m & 1))) & 0xffffu);
        track16(ovf, v);
        store16(tt ? tt + 4 * ls + 32 * m : nullptr, v, sc);
        split16<false>(v, ah[m], al[m]);
    }
    TMK_LOAD(5, mk) layer16_tan(w, blk, lane, ah, al, bh, bl, mk, tt ? tt + 5 * ls : nullptr, sc, ovf);
    if (!colsum6) {
        TMK_LOAD(6, mk) layer16_tan(w, blk, lane, bh, bl, ah, al, mk, tt ? tt + 6 * ls : nullptr, sc, ovf);
        if (!(fmaxf(ovf, __shfl_xor(ovf, 32)) < F16_RANGE) && tt) {      // (false for inf and for the NaN an inf times 0 leaves)
            zero_train_rows(tt, ls, 7);
            if (half == 0 && range_count) atomicAdd(range_count, 1);
        }
        return;
    }
    float h6[8][16];
    TMK_LOAD(6, mk) layer16_tan(w, blk, lane, bh, bl, ah, al, mk, nullptr, sc, ovf, h6);
#undef TMK_LOAD
    const bool bad = !(fmaxf(ovf, __shfl_xor(ovf, 32)) < F16_RANGE);
    if (bad && tt) {
        zero_train_rows(tt, ls, 6);
        if (half == 0 && range_count) atomicAdd(range_count, 1);
    }
    // column sums of the block's 128 rows: a sample that is not listed (tail of the last block) or left the fp16 range counts as
    // zeros, like its stored rows; lanes -> half-wave sums (DPP) -> the four waves through LDS -> one atomic per feature and block
    float* const s_sum = reinterpret_cast<float*>(&s_pe[0][0]);      // [4][256] (the encoding's operand slots are dead by now)
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float t = half_wave_sum_to_last((valid && !bad) ? h6[m][r] : 0.0f);
            // register r of block m, half h: feature 32 m + 8 (r / 4) + 4 h + r % 4 (the stores' address map)
            if ((lane & 31) == 31) s_sum[wave * 256 + 32 * m + 8 * (r >> 2) + 4 * half + (r & 3)] = t;
        }
    __syncthreads();
    atomicAdd(colsum6 + tid, (s_sum[tid] + s_sum[256 + tid]) + (s_sum[512 + tid] + s_sum[768 + tid]));
}

void dsn_launch_tangent16(const float* packed, const float* x_c, const float* u, int64_t N, const void* masks, float* tr_t,
                          float* gmax, hipStream_t st, int32_t* range_count, const int32_t* row_list, const int32_t* row_count,
                          float* colsum6) {
    int64_t blocks = (N + 127) / 128;
    if (blocks == 0) return;
    hipLaunchKernelGGL(k_tangent16, dim3((unsigned)blocks), dim3(F16_THREADS), 0, st, packed, x_c, u, N, (const uint4*)masks, tr_t,
                       (uint32_t*)gmax, range_count, row_list, row_count, colsum6);
}

// ---------------------------------------------------------------------------------------------
// k_adjoint16 : the ADJOINT pass of the training backward below its seed: given ahat_6 (the masked adjoint of
// dL/dsigma * sigma + dL/dessence . essence at stage2.4, row-major [N,256]) it runs the transposed layers with the recorded
// relu patterns and stores ahat_5 ... ahat_0.  Linear in the seed, so every sample runs on seed / max|seed| and its outputs
// are stored times that maximum.  Same weight images and reverse flavour of dense16 as k_field16<reverse>.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(F16_THREADS, 1)
k_adjoint16(const float* __restrict__ packed, int64_t N, const uint4* __restrict__ masks, const float* __restrict__ a_in,
            float* __restrict__ tr_a, uint32_t* __restrict__ gmax, int32_t* __restrict__ range_count,
            const int32_t* __restrict__ row_list, const int32_t* __restrict__ row_count) {
    DSN_OWN_SIMD_T(2);
    __shared__ __attribute__((aligned(16))) char ring[F16_RING_SLOTS * 4096];
    __shared__ uint32_t s_mask[7][4][F16_THREADS];
    const int tid = threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5;
    const int64_t NL = row_count ? (int64_t)(*row_count) : N;
    if ((int64_t)blockIdx.x * 128 >= NL) return;      // block-uniform
    int64_t pt = ((int64_t)blockIdx.x * 4 + wave) * 32 + (lane & 31);
    const bool valid = pt < NL;
    if (!valid) pt = NL - 1;
    if (row_list) pt = (int64_t)row_list[pt];
    {
        const uint4* mrec = masks + ((size_t)pt * 2 + half) * 7;
#pragma unroll
        for (int L = 0; L < 6; ++L) {
            const uint4 q = mrec[L];
            s_mask[L][0][tid] = q.x; s_mask[L][1][tid] = q.y; s_mask[L][2][tid] = q.z; s_mask[L][3][tid] = q.w;
        }
    }
    // the seed, row-major -> accumulator layout, and its magnitude
    const float* in = a_in + pt * 256 + 4 * half;
    f32x16 v[8];
    float sc = 0.0f;
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 x = *reinterpret_cast<const float4*>(in + 32 * m + 8 * q);
            v[m][4 * q] = x.x; v[m][4 * q + 1] = x.y; v[m][4 * q + 2] = x.z; v[m][4 * q + 3] = x.w;
            sc = fmaxf(sc, fmaxf(fmaxf(fabsf(x.x), fabsf(x.y)), fmaxf(fabsf(x.z), fabsf(x.w))));
        }
    sc = fmaxf(sc, __shfl_xor(sc, 32));
    if (!(sc > 1e-30f) || !(sc < 3.0e38f)) sc = 0.0f;       // (a denormal seed: 1 / max|seed| would be inf - see k_tangent16)
    const float inv = sc > 0.0f ? 1.0f / sc : 0.0f;
    if (gmax) {
        float wm = valid ? sc : 0.0f;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) wm = fmaxf(wm, __shfl_xor(wm, o));
        if (lane == 0 && __float_as_uint(wm) > __atomic_load_n(gmax, __ATOMIC_RELAXED)) atomicMax(gmax, __float_as_uint(wm));
    }
    half8 ah[8][2], al[8][2], bh[8][2], bl[8][2];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[m][r] = sc > 0.0f ? v[m][r] * inv : 0.0f;
        split16<true>(v[m], ah[m], al[m]);
    }
    W16 w;
    w.g = reinterpret_cast<const char*>(packed + OFF16_BASE) + wave * 8192 + 4096 + lane * 16;
    w.ring = ring;
    w.ring_off = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)ring;
    w.wave = wave;
    w.spread = false;
    w16_begin(w, lane, F16_FIRST_BWD_BLOCK);
    int blk = F16_FIRST_BWD_BLOCK;
    const int64_t ls = N * 256;
    float* const ta = valid ? tr_a + pt * 256 + 4 * half : nullptr;
#define AMK_LOAD(L, mk) { mk[0] = s_mask[L][0][tid]; mk[1] = s_mask[L][1][tid]; mk[2] = s_mask[L][2][tid]; mk[3] = s_mask[L][3][tid]; }
    uint32_t mk[4];
    float ovf = 0.0f;          // range guard: the seed is normalised, what the six transposed layers make of it is not bounded
    AMK_LOAD(5, mk) layer16_bwd<true>(w, blk, lane, ah, al, bh, bl, mk, ovf, ta ? ta + 5 * ls : nullptr, sc);
    AMK_LOAD(4, mk) layer16_bwd<true>(w, blk, lane, bh, bl, ah, al, mk, ovf, ta ? ta + 4 * ls : nullptr, sc);
    AMK_LOAD(3, mk)
#pragma unroll
    for (int m = 0; m < 8; ++m) {       // stage2.0^T, h part
        f32x16 aM = zero16(), aC = zero16();
        dense16<8, true>(w, blk, lane, ah, al, aM, aC);
        f32x16 x = fold16(aM, aC);
        mask16(x, (mk[m >> 1] >> (16 * (m & 1))) & 0xffffu);
        track16(ovf, x);
        store16(ta ? ta + 3 * ls + 32 * m : nullptr, x, sc);
        split16<true>(x, bh[m], bl[m]);
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {       // its positional-encoding rows: not needed here, consumed to stay on the stream
        f32x16 aM = zero16(), aC = zero16();
        dense16<8, true>(w, blk, lane, ah, al, aM, aC);
    }
    AMK_LOAD(2, mk) layer16_bwd<true>(w, blk, lane, bh, bl, ah, al, mk, ovf, ta ? ta + 2 * ls : nullptr, sc);
    AMK_LOAD(1, mk) layer16_bwd<true>(w, blk, lane, ah, al, bh, bl, mk, ovf, ta ? ta + 1 * ls : nullptr, sc);
    AMK_LOAD(0, mk) layer16_bwd<true>(w, blk, lane, bh, bl, ah, al, mk, ovf, ta ? ta + 0 * ls : nullptr, sc);
#undef AMK_LOAD
    if (!(fmaxf(ovf, __shfl_xor(ovf, 32)) < F16_RANGE) && ta) {
        zero_train_rows(ta, ls, 6);
        if (half == 0 && range_count) atomicAdd(range_count, 1);
    }
}

void dsn_launch_adjoint16(const float* packed, int64_t N, const void* masks, const float* a_in, float* tr_a, float* gmax,
                          hipStream_t st, int32_t* range_count, const int32_t* row_list, const int32_t* row_count) {
    int64_t blocks = (N + 127) / 128;
    if (blocks == 0) return;
    hipLaunchKernelGGL(k_adjoint16, dim3((unsigned)blocks), dim3(F16_THREADS), 0, st, packed, N, (const uint4*)masks, a_in, tr_a,
                       (uint32_t*)gmax, range_count, row_list, row_count);
}

// ---------------------------------------------------------------------------------------------
// k_screen16 : eval-mode density screen.  71 % of the non-transparent samples of a frame end up with sigma <= 0 and
// contribute exactly nothing (alpha = 1 - exp(-relu(sigma) dist) = 0); the accurate forward pass is only needed to
// KNOW that.  This kernel evaluates the trunk with the hi halves alone (plain fp16 operands, fp32 accumulation: one
// MFMA product instead of three, half the weight traffic, no residuals, no relu records, no colour head) and keeps,
// next to sigma~, the magnitude S1 = |b_d| + sum_i |w_d,i h6_i| of what was added up.  A sample is declared empty when
//     sigma~ < -(F16_SCREEN_REL * S1 + F16_SCREEN_ABS),
// a margin 36x the largest fp16-vs-fp32 deviation measured on the benchmark frame relative to S1 (2.75e-4 S1)
// (tests/test_gpu_render.py::test_density_screen_margin); every other sample goes to the accurate pass.  An empty
// sample keeps sigma~ (< 0) as its density, so the compositor sees the same exact zero.  off unless DSN_DENSITY_SCREEN is given.
// ---------------------------------------------------------------------------------------------
#ifndef F16_SCREEN_SGB
#define F16_SCREEN_SGB 0      // n > 0: pin 1 MFMA : n VALU inside every block of the screen kernel
#endif
#ifndef F16_SCREEN_ACC
#define F16_SCREEN_ACC 1
#endif
#define F16_SCREEN_REL DSN_SCREEN_MARGIN_DEFAULT
#define F16_SCREEN_ABS DSN_SCREEN_MARGIN_DEFAULT
#define F16_SCREEN_BLOCKS (OFF_RGB1 / DSN_BLK)       // 416: stage1.0 ... stage2.4

// hi quarters only: quarter q = 0 (k-step 0) and 2 (k-step 1) of every block; the 16 pieces of a chunk are shared out as
// (quarter, half of the chunk's blocks) per wave - 4 contiguous KB each
template <int NW>
__device__ __forceinline__ void w16s_stage(const W16& w, int c) {
    const char* src = w.g + (size_t)c * 32768;
    // compact ring of the screen: [chunk parity][k-step][block in chunk][1 KB] = 2 x 16 KB; NW = 4: 4 KB per wave, 8: 2 KB
    const unsigned dst = w.ring_off + (c & 1) * 16384 + (w.wave & 1) * 8192 + (w.wave >> 1) * (16384 / NW);
    if (NW == 4)
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %0, off\n\tglobal_load_lds_dwordx4 %0, off offset:1024\n\t"
                     "global_load_lds_dwordx4 %0, off offset:2048\n\tglobal_load_lds_dwordx4 %0, off offset:3072"
                     : : "v"(src), "s"(dst) : "memory", "m0");
    else
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %0, off\n\tglobal_load_lds_dwordx4 %0, off offset:1024"
                     : : "v"(src), "s"(dst) : "memory", "m0");
}
#ifndef F16_SABL
#define F16_SABL 0            // screen timing ablations (WRONG results): 1 no ring barrier, 4 no LDS-DMA, 16 empty epilogue slices
#endif
template <int NW>
__device__ __forceinline__ void w16s_boundary(const W16& w, int b) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if !(F16_SABL & 1)
    __syncthreads();
#endif
    const int c = b / F16_CHUNK;
#if !(F16_SABL & 4)
    if (c + 1 < F16_SCREEN_BLOCKS / F16_CHUNK) w16s_stage<NW>(w, c + 1);
#endif
}
__device__ __forceinline__ void w16s_read(const W16& w, int b, int lane, half8& h0, half8& h1) {
    const char* s = w.ring + ((b >> 3) & 1) * 16384 + (b & 7) * 1024 + lane * 16;
    h0 = *reinterpret_cast<const half8*>(s);
    h1 = *reinterpret_cast<const half8*>(s + 8192);
}
// a0 += 64 Wh[32 rows][32*KB k] xh.  F16_SCREEN_ACC = 1: one accumulator (the second wave of the SIMD covers the
// dependent MFMA pair, and the epilogue saves an accumulator read and an add per element); 2: one per k-step
template <int NW, int KB, class Hook = NoHook>
__device__ __forceinline__ void dense16s(W16& w, int& blk, int lane, const half8 (&xh)[KB][2], f32x16& a0, f32x16& a1,
                                         Hook&& hook = NoHook()) {
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        half8 n0 = w.h0, n1 = w.h1;
        if (blk + 1 < F16_SCREEN_BLOCKS) {
            if (((blk + 1) & (F16_CHUNK - 1)) == 0) w16s_boundary<NW>(w, blk + 1);
            w16s_read(w, blk + 1, lane, n0, n1);
        }
        a0 = MFMA16(w.h0, xh[kb][0], a0);
        if (F16_SCREEN_ACC == 1) a0 = MFMA16(w.h1, xh[kb][1], a0);
        else a1 = MFMA16(w.h1, xh[kb][1], a1);
        hook(kb);
        w.h0 = n0; w.h1 = n1;
        ++blk;
#if F16_SCREEN_SGB
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, F16_SCREEN_SGB, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#endif
    }
}
// two accumulator elements (64 z) -> relu(z) as packed fp16: convert first, then packed max and packed scale by 2^-6
// (exact; a |64 z| beyond the fp16 range becomes inf and the sample is simply kept for the accurate pass)
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
// `ovf` (range guard): running packed maximum of the activations.  An activation beyond the fp16 range is +inf here, and an
// inf that meets a negative weight becomes -inf and then 0 in the next ReLU - a FINITE, wrong sigma~; so the overflow is
// remembered and such a sample is kept for the accurate pass whatever its sigma~ says.
__device__ __forceinline__ half2v relu_pair16(float u, float v, half2v& ovf) {
    half2v h = {(_Float16)u, (_Float16)v};
    const half2v z = {(_Float16)0.0f, (_Float16)0.0f};
    const half2v s = {(_Float16)F16_FWD_INV, (_Float16)F16_FWD_INV};
    h = __builtin_elementwise_max(h, z);
    // before the exact 2^-6 scaling: 64 z is what has to fit.  (asm: as a builtin the running maximum is re-associated into one
    // tree at the end of the tile and every h stays live until then)
    asm("v_pk_max_f16 %0, %0, %1" : "+v"(ovf) : "v"(h));
    return h * s;
}
__device__ __forceinline__ void relu_half16(const f32x16& a, const f32x16& b, half8 (&y)[2], half2v& ovf) {
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        const half2v h = F16_SCREEN_ACC == 1 ? relu_pair16(a[r], a[r + 1], ovf) : relu_pair16(a[r] + b[r], a[r + 1] + b[r + 1], ovf);
        y[r >> 3][r & 7] = h[0];
        y[r >> 3][(r & 7) + 1] = h[1];
    }
}
template <int NW>
__device__ __forceinline__ void layer16s(W16& w, int& blk, int lane, const float* __restrict__ bias, const half8 (&xh)[8][2],
                                         half8 (&yh)[8][2], half2v& ovf) {
    const int half = lane >> 5;
    f32x16 p0 = zero16(), p1 = zero16();
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        f32x16 a0 = rows16(bias, m, half), a1 = zero16();
        if (m == 0) dense16s<NW, 8>(w, blk, lane, xh, a0, a1);
        else dense16s<NW, 8>(w, blk, lane, xh, a0, a1, [&](int kb) {
            const int r = 2 * kb;
#if F16_SABL & 16
            asm volatile("" : : "v"(p0[r]), "v"(p0[r + 1]));
            if (kb == 0) yh[m - 1][0] = xh[m - 1][0];
            return;
#endif
            const half2v h = F16_SCREEN_ACC == 1 ? relu_pair16(p0[r], p0[r + 1], ovf) : relu_pair16(p0[r] + p1[r], p0[r + 1] + p1[r + 1], ovf);
            yh[m - 1][r >> 3][r & 7] = h[0];
            yh[m - 1][r >> 3][(r & 7) + 1] = h[1];
        });
        p0 = a0; p1 = a1;
    }
    relu_half16(p0, p1, yh[7], ovf);
}

// NW = 4: 256 threads, two workgroups per CU (58 KB of LDS each).  NW = 8: ONE workgroup of 512 threads per CU - the same eight
// waves per CU share one weight ring, i.e. half the L2 -> LDS weight traffic per sample (DSN_SCREEN_WAVES selects).
template <int NW>
__global__ void __launch_bounds__(64 * NW, 1)
k_screen16(const float* __restrict__ packed, const DsnFrameState* __restrict__ fs, const float* __restrict__ x_c, int64_t N,
           const int32_t* __restrict__ active_list, const int32_t* __restrict__ active_count, float* __restrict__ sigma,
           int32_t* __restrict__ keep_list, int32_t* __restrict__ keep_count, float* __restrict__ dbg_sigma,
           float* __restrict__ dbg_s1, float margin_override, int32_t* __restrict__ audit_list, int32_t* __restrict__ audit_count,
           int audit_cap) {
    __shared__ __attribute__((aligned(16))) char ring[2 * 16384];
    __shared__ __attribute__((aligned(16))) float s_vec[256 + 2304 + 8];
    __shared__ __attribute__((aligned(16))) half8 s_pe[4][64 * NW];
    __shared__ int s_cnt[NW];
    __shared__ int s_base;
#ifndef F16_SHARE_SIMD
    // no other kernel's wave beside these (dsn_common.h, DSN_OWN_SIMD): the 512-thread form keeps TWO of its own waves per SIMD, 256
    // registers each; the 256-thread form one
    if (NW == 8) asm volatile("v_mov_b32 v255, 0" ::: "v255"); else DSN_OWN_SIMD();
#endif
    const int tid0 = threadIdx.x;
    const int lane0 = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int tid = tid0, lane = lane0;
    const int64_t count = active_list ? (int64_t)(*active_count) : N;
    // persistent workgroups: tile t = samples [t * 32 NW, (t + 1) * 32 NW) of the list, tiles dealt round-robin.  The small vectors
    // are fetched once per workgroup, the next tile's points and the next tile's first weight chunk travel under the current tile
    const int64_t ntiles = (count + 32 * NW - 1) / (32 * NW);
    if ((int64_t)blockIdx.x >= ntiles) return;
    for (int i = tid; i < 256 + 2304 + 8; i += 64 * NW)
        s_vec[i] = i < 256 ? fs->bias0[i] : (i < 2560 ? packed[OFF_B1 + (i - 256)] : packed[OFF_SCAL + (i - 2560)]);
    for (int i = tid; i < 256 + (OFF_WDEN - OFF_B1); i += 64 * NW) s_vec[i] *= F16_FWD_SCALE;
    const float* const v_b1 = s_vec + 256;
    const float* const v_wden = s_vec + 256 + (OFF_WDEN - OFF_B1);
    W16 w;
    w.g = reinterpret_cast<const char*>(packed + OFF16_BASE) + (wave & 1) * 16384 + (wave >> 1) * (16384 / NW) + lane * 16;
    w.ring = ring;
    w.ring_off = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)ring;
    w.wave = wave;
    w16s_stage<NW>(w, 0);
    auto tile_point = [&](int64_t t, bool& ok) -> int64_t {
        int64_t sl = (t * NW + wave) * 32 + (lane & 31);
        ok = sl < count;
        if (!ok) sl = count - 1;
        return active_list ? (int64_t)active_list[sl] : sl;
    };
    bool valid_n;
    int64_t pt_n = tile_point(blockIdx.x, valid_n);
    float xn[3] = {x_c[3 * pt_n], x_c[3 * pt_n + 1], x_c[3 * pt_n + 2]};
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    // (opaque copies: the unrolled body holds hundreds of LDS addresses derived from these two; loop-invariant, the compiler
    //  would hoist them all out of the tile loop and spill)
    tid = tid0; lane = lane0;
    asm volatile("" : "+v"(tid), "+v"(lane), "+v"(w.g));      // (w.g: one 64-bit DMA source address per chunk otherwise)
    const int half = lane >> 5;
    const bool valid = valid_n;
    const int64_t pt = pt_n;
    const float xa[3] = {xn[0], xn[1], xn[2]};
    const bool more = tile + gridDim.x < ntiles;      // workgroup-uniform
    if (more) pt_n = tile_point(tile + gridDim.x, valid_n);       // (its x_c follows further down, when the index has arrived)
    w16s_boundary<NW>(w, 0);
    w16s_read(w, 0, lane, w.h0, w.h1);
    int blk = 0;
    half8 ah[8][2], bh[8][2];
    half8 ph[2][2];
    half2v ovf = {(_Float16)0.0f, (_Float16)0.0f};
    {
        f32x16 pe[2];
#pragma unroll
        for (int t = 0; t < 30; ++t) {
            float s, c;
            dsn_sincos(xa[t % 3] * (float)(1 << (t / 3)), s, c);
            pe[t >> 4][t & 15] = half ? c : s;
        }
        pe[1][14] = half ? xa[1] : xa[0];
        pe[1][15] = half ? 0.0f : xa[2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) ph[b][r >> 3][r & 7] = (_Float16)pe[b][r];
        s_pe[0][tid] = ph[0][0]; s_pe[1][tid] = ph[0][1]; s_pe[2][tid] = ph[1][0]; s_pe[3][tid] = ph[1][1];
    }
    // stage1.0
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        f32x16 a0 = rows16(s_vec, m, half), a1 = zero16();
        dense16s<NW, 2>(w, blk, lane, ph, a0, a1);
        relu_half16(a0, a1, ah[m], ovf);
    }
    layer16s<NW>(w, blk, lane, v_b1 + 0 * 256, ah, bh, ovf);
    if (more) { xn[0] = x_c[3 * pt_n]; xn[1] = x_c[3 * pt_n + 1]; xn[2] = x_c[3 * pt_n + 2]; }
    layer16s<NW>(w, blk, lane, v_b1 + 1 * 256, bh, ah, ovf);
    layer16s<NW>(w, blk, lane, v_b1 + 2 * 256, ah, bh, ovf);
    // stage2.0 : [h, pe] -> 256
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        f32x16 a0 = rows16(v_b1 + 3 * 256, m, half), a1 = zero16();
        dense16s<NW, 8>(w, blk, lane, bh, a0, a1);
        half8 qh[2][2];
        qh[0][0] = s_pe[0][tid]; qh[0][1] = s_pe[1][tid]; qh[1][0] = s_pe[2][tid]; qh[1][1] = s_pe[3][tid];
        dense16s<NW, 2>(w, blk, lane, qh, a0, a1);
        relu_half16(a0, a1, ah[m], ovf);
    }
    layer16s<NW>(w, blk, lane, v_b1 + 4 * 256, ah, bh, ovf);
    // stage2.4 + density head: sigma~ and the magnitude of its terms
    float sg = 0.0f, s1 = 0.0f;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        f32x16 a0 = rows16(v_b1 + 5 * 256, m, half), a1 = zero16();
        dense16s<NW, 8>(w, blk, lane, bh, a0, a1);
        const f32x16 wd = rows16(v_wden, m, half);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float h = fmaxf((F16_SCREEN_ACC == 1 ? a0[r] : a0[r] + a1[r]) * F16_FWD_INV, 0.0f);
            const float term = wd[r] * h;
            sg += term;
            s1 += fabsf(term);
        }
    }
    sg += __shfl_xor(sg, 32);
    s1 += __shfl_xor(s1, 32);
    const float bd = s_vec[2560];
    sg += bd;
    s1 += fabsf(bd);
    // range guard: the last layer's activations are fp32 here, every earlier one went through `ovf`
    float omax = fmaxf((float)ovf[0], (float)ovf[1]);
    omax = fmaxf(omax, __shfl_xor(omax, 32));
    const bool in_range = omax < F16_RANGE;              // false for inf (and for the NaN an inf times 0 leaves)
    const bool mine = valid && half == 0;
    // margin: packed[OFF_SCAL + 5] - the conservative default written by dsn_pack_params (F16_SCREEN_REL) or the value
    // dsn_calibrate_screen measured for THESE parameters (10x the largest deviation seen, +inf = never declare anything
    // empty); DSN_SCREEN_MARGIN (experiments) overrides it
    const float margin = margin_override > 0.0f ? margin_override : s_vec[2560 + 5];
    const bool empty = in_range && sg < -(margin * s1 + margin);
    if (mine) {
        if (empty) sigma[pt] = sg;
        if (dbg_sigma) { dbg_sigma[pt] = in_range ? sg : dsn_nan_flag(); dbg_s1[pt] = in_range ? s1 : dsn_nan_flag(); }
    }
    // audit (DSN_SCREEN_AUDIT): a pseudo-random 1/128 of the samples declared empty go through the accurate pass anyway
    // and are remembered; k_screen_audit then counts those whose accurate density is positive (there must be none)
    bool audit = false;
    if (audit_list && mine && empty) {
        uint32_t hsh = (uint32_t)pt * 2654435761u;
        hsh ^= hsh >> 15;
        if ((hsh & 127u) == 5u) {
            const int a = atomicAdd(audit_count, 1);
            if (a < audit_cap) { audit_list[a] = (int32_t)pt; audit = true; }
        }
    }
    // the others go to the accurate pass: workgroup-aggregated append
    const bool keep = mine && (!empty || audit);
    const unsigned long long bm = __ballot(keep);
    if (lane == 0) s_cnt[wave] = __popcll(bm);
    __syncthreads();
    // (every wave has read its last weight block: the ring is free for the next tile's first chunk)
    if (more) w16s_stage<NW>(w, 0);
    if (tid == 0) {
        int tot = 0;
#pragma unroll
        for (int k = 0; k < NW; ++k) tot += s_cnt[k];
        s_base = tot ? atomicAdd(keep_count, tot) : 0;
    }
    __syncthreads();
    if (keep) {
        int off = s_base + __popcll(bm & ((1ull << lane) - 1ull));
        for (int k = 0; k < wave; ++k) off += s_cnt[k];
        keep_list[off] = (int32_t)pt;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// k_screen16x2 : the same screen with TWO 32-sample tiles per wave (4 waves x 64 samples = the same 256-sample workgroup tile).
// Every weight operand read from LDS now feeds two MFMAs (one per tile) instead of one: half the LDS operand bytes and half the
// LDS-DMA instructions per sample, four waves instead of eight at the ring barrier, and the two tiles' accumulators alternate,
// so no MFMA waits for its predecessor.  Per sample the sequence of products and roundings is the one of k_screen16 (its own
// accumulator takes k-step 0 then k-step 1 of every block): sigma~ and S1 are bit-identical.  One wave per SIMD (the activations
// of two tiles fill the registers), persistent workgroups.  MEASURED SLOWER than k_screen16<8> (see dsn_launch_screen16): kept as
// the DSN_SCREEN_WAVES=2 experiment, not the default.
// ---------------------------------------------------------------------------------------------
template <int I>
__device__ __forceinline__ void w16x_stage_part(const W16& w, int c) {      // piece I of 4: wave's (k-step, half of the blocks) share
    const char* src = w.g + (size_t)c * 32768;
    const unsigned dst = w.ring_off + (c & 1) * 16384 + (w.wave & 1) * 8192 + (w.wave >> 1) * 4096;
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off offset:%2" : : "v"(src), "s"(dst), "n"(1024 * I) : "memory", "m0");
}
__device__ __forceinline__ void w16x_boundary(const W16& w) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}
template <int KB, class Hook = NoHook>
__device__ __forceinline__ void dense16x(W16& w, int& blk, int lane, const half8 (&x0)[KB][2], const half8 (&x1)[KB][2], f32x16& a0,
                                         f32x16& a1, Hook&& hook = NoHook()) {
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        half8 n0 = w.h0, n1 = w.h1;
        if (blk + 1 < F16_SCREEN_BLOCKS) {
            if (((blk + 1) & (F16_CHUNK - 1)) == 0) w16x_boundary(w);
            {   // the chunk after the one block blk + 1 lives in: one 1 KB piece behind each of its first four blocks
                const int pos = (blk + 1) & (F16_CHUNK - 1), cn = (blk + 1) / F16_CHUNK + 1;
                if (cn < F16_SCREEN_BLOCKS / F16_CHUNK) {
                    if (pos == 0) w16x_stage_part<0>(w, cn);
                    if (pos == 1) w16x_stage_part<1>(w, cn);
                    if (pos == 2) w16x_stage_part<2>(w, cn);
                    if (pos == 3) w16x_stage_part<3>(w, cn);
                }
            }
            w16s_read(w, blk + 1, lane, n0, n1);
        }
        a0 = MFMA16(w.h0, x0[kb][0], a0);
        a1 = MFMA16(w.h0, x1[kb][0], a1);
        a0 = MFMA16(w.h1, x0[kb][1], a0);
        a1 = MFMA16(w.h1, x1[kb][1], a1);
        hook(kb);
        w.h0 = n0; w.h1 = n1;
        ++blk;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}
__device__ __forceinline__ void relu_slice16x(const f32x16& p, int kb, half8 (&y)[2], half2v& ovf) {
    const int r = 2 * kb;
    const half2v h = relu_pair16(p[r], p[r + 1], ovf);
    y[r >> 3][r & 7] = h[0];
    y[r >> 3][(r & 7) + 1] = h[1];
}
__device__ __forceinline__ void layer16x(W16& w, int& blk, int lane, const float* __restrict__ bias, const half8 (&x0)[8][2],
                                         const half8 (&x1)[8][2], half8 (&y0)[8][2], half8 (&y1)[8][2], half2v& ovf) {
    const int half = lane >> 5;
    f32x16 p0 = zero16(), p1 = zero16();
    f32x16 bnext = rows16(bias, 0, half);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        f32x16 a0 = bnext, a1 = bnext;
        if (m + 1 < 8) bnext = rows16(bias, m + 1, half);
        if (m == 0) dense16x<8>(w, blk, lane, x0, x1, a0, a1);
        else dense16x<8>(w, blk, lane, x0, x1, a0, a1, [&](int kb) {
            relu_slice16x(p0, kb, y0[m - 1], ovf);
            relu_slice16x(p1, kb, y1[m - 1], ovf);
        });
        p0 = a0; p1 = a1;
    }
    f32x16 z = zero16();
    relu_half16(p0, z, y0[7], ovf);
    relu_half16(p1, z, y1[7], ovf);
}
__global__ void __launch_bounds__(256, 1)
k_screen16x2(const float* __restrict__ packed, const DsnFrameState* __restrict__ fs, const float* __restrict__ x_c, int64_t N,
             const int32_t* __restrict__ active_list, const int32_t* __restrict__ active_count, float* __restrict__ sigma,
             int32_t* __restrict__ keep_list, int32_t* __restrict__ keep_count, float* __restrict__ dbg_sigma,
             float* __restrict__ dbg_s1, float margin_override, int32_t* __restrict__ audit_list, int32_t* __restrict__ audit_count,
             int audit_cap) {
    static_assert(F16_SCREEN_ACC == 1, "k_screen16x2 keeps one accumulator per tile");
    __shared__ __attribute__((aligned(16))) char ring[2 * 16384];
    __shared__ __attribute__((aligned(16))) float s_vec[256 + 2304 + 8];
    __shared__ __attribute__((aligned(16))) half8 s_pe[2][4][256];
    __shared__ int s_cnt[4];
    __shared__ int s_base;
#ifndef F16_SHARE_SIMD
    DSN_OWN_SIMD();
#endif
    const int tid0 = threadIdx.x;
    const int lane0 = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t count = active_list ? (int64_t)(*active_count) : N;
    const int64_t ntiles = (count + 255) / 256;
    if ((int64_t)blockIdx.x >= ntiles) return;
    for (int i = tid0; i < 256 + 2304 + 8; i += 256)
        s_vec[i] = i < 256 ? fs->bias0[i] : (i < 2560 ? packed[OFF_B1 + (i - 256)] : packed[OFF_SCAL + (i - 2560)]);
    for (int i = tid0; i < 256 + (OFF_WDEN - OFF_B1); i += 256) s_vec[i] *= F16_FWD_SCALE;
    const float* const v_b1 = s_vec + 256;
    const float* const v_wden = s_vec + 256 + (OFF_WDEN - OFF_B1);
    W16 w;
    w.g = reinterpret_cast<const char*>(packed + OFF16_BASE) + (wave & 1) * 16384 + (wave >> 1) * 4096 + lane0 * 16;
    w.ring = ring;
    w.ring_off = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)ring;
    w.wave = wave;
    w16s_stage<4>(w, 0);
    // wave `wave` of tile t owns list slots (4 t + wave) * 64 ... + 63: lane & 31 of its tile 0, + 32 of its tile 1
    auto tile_point = [&](int64_t t, int which, bool& ok) -> int64_t {
        int64_t sl = (t * 4 + wave) * 64 + 32 * which + (lane0 & 31);
        ok = sl < count;
        if (!ok) sl = count - 1;
        return active_list ? (int64_t)active_list[sl] : sl;
    };
    bool valid_n[2];
    int64_t pt_n[2];
    float xn[2][3];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        pt_n[q] = tile_point(blockIdx.x, q, valid_n[q]);
        xn[q][0] = x_c[3 * pt_n[q]]; xn[q][1] = x_c[3 * pt_n[q] + 1]; xn[q][2] = x_c[3 * pt_n[q] + 2];
    }
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int tid = tid0, lane = lane0;
    asm volatile("" : "+v"(tid), "+v"(lane), "+v"(w.g));      // (opaque copies: see k_field16)
    const int half = lane >> 5;
    const bool valid[2] = {valid_n[0], valid_n[1]};
    const int64_t pt[2] = {pt_n[0], pt_n[1]};
    const float xa[2][3] = {{xn[0][0], xn[0][1], xn[0][2]}, {xn[1][0], xn[1][1], xn[1][2]}};
    const bool more = tile + gridDim.x < ntiles;      // workgroup-uniform
    if (more) { pt_n[0] = tile_point(tile + gridDim.x, 0, valid_n[0]); pt_n[1] = tile_point(tile + gridDim.x, 1, valid_n[1]); }
    w16x_boundary(w);
    w16x_stage_part<0>(w, 1);
    w16s_read(w, 0, lane, w.h0, w.h1);
    int blk = 0;
    half8 ah0[8][2], ah1[8][2], bh0[8][2], bh1[8][2];
    half8 ph0[2][2], ph1[2][2];
    half2v ovf = {(_Float16)0.0f, (_Float16)0.0f};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        f32x16 pe[2];
#pragma unroll
        for (int t = 0; t < 30; ++t) {
            float sn, cs;
            dsn_sincos(xa[q][t % 3] * (float)(1 << (t / 3)), sn, cs);
            pe[t >> 4][t & 15] = half ? cs : sn;
        }
        pe[1][14] = half ? xa[q][1] : xa[q][0];
        pe[1][15] = half ? 0.0f : xa[q][2];
        half8 (&ph)[2][2] = q ? ph1 : ph0;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) ph[b][r >> 3][r & 7] = (_Float16)pe[b][r];
        s_pe[q][0][tid] = ph[0][0]; s_pe[q][1][tid] = ph[0][1]; s_pe[q][2][tid] = ph[1][0]; s_pe[q][3][tid] = ph[1][1];
    }
    f32x16 zz = zero16();
    // stage1.0
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        f32x16 a0 = rows16(s_vec, m, half), a1 = a0;
        dense16x<2>(w, blk, lane, ph0, ph1, a0, a1);
        relu_half16(a0, zz, ah0[m], ovf);
        relu_half16(a1, zz, ah1[m], ovf);
    }
    layer16x(w, blk, lane, v_b1 + 0 * 256, ah0, ah1, bh0, bh1, ovf);
    if (more) {
#pragma unroll
        for (int q = 0; q < 2; ++q) { xn[q][0] = x_c[3 * pt_n[q]]; xn[q][1] = x_c[3 * pt_n[q] + 1]; xn[q][2] = x_c[3 * pt_n[q] + 2]; }
    }
    layer16x(w, blk, lane, v_b1 + 1 * 256, bh0, bh1, ah0, ah1, ovf);
    layer16x(w, blk, lane, v_b1 + 2 * 256, ah0, ah1, bh0, bh1, ovf);
    // stage2.0 : [h, pe] -> 256
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        f32x16 a0 = rows16(v_b1 + 3 * 256, m, half), a1 = a0;
        dense16x<8>(w, blk, lane, bh0, bh1, a0, a1);
        half8 q0[2][2], q1[2][2];
        q0[0][0] = s_pe[0][0][tid]; q0[0][1] = s_pe[0][1][tid]; q0[1][0] = s_pe[0][2][tid]; q0[1][1] = s_pe[0][3][tid];
        q1[0][0] = s_pe[1][0][tid]; q1[0][1] = s_pe[1][1][tid]; q1[1][0] = s_pe[1][2][tid]; q1[1][1] = s_pe[1][3][tid];
        dense16x<2>(w, blk, lane, q0, q1, a0, a1);
        relu_half16(a0, zz, ah0[m], ovf);
        relu_half16(a1, zz, ah1[m], ovf);
    }
    layer16x(w, blk, lane, v_b1 + 4 * 256, ah0, ah1, bh0, bh1, ovf);
    // stage2.4 + density head: sigma~ and the magnitude of its terms, per tile
    float sgv[2] = {0.0f, 0.0f}, s1v[2] = {0.0f, 0.0f};
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        f32x16 a0 = rows16(v_b1 + 5 * 256, m, half), a1 = a0;
        dense16x<8>(w, blk, lane, bh0, bh1, a0, a1);
        const f32x16 wd = rows16(v_wden, m, half);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float h0 = fmaxf(a0[r] * F16_FWD_INV, 0.0f), h1 = fmaxf(a1[r] * F16_FWD_INV, 0.0f);
            const float t0 = wd[r] * h0, t1 = wd[r] * h1;
            sgv[0] += t0; s1v[0] += fabsf(t0);
            sgv[1] += t1; s1v[1] += fabsf(t1);
        }
    }
    // range guard: the last layer's activations are fp32 here, every earlier one went through `ovf` (shared by the lane's two
    // samples: an overflow in either keeps both for the accurate pass - conservative)
    float omax = fmaxf((float)ovf[0], (float)ovf[1]);
    omax = fmaxf(omax, __shfl_xor(omax, 32));
    const bool in_range = omax < F16_RANGE;
    const float bd = s_vec[2560];
    const float margin = margin_override > 0.0f ? margin_override : s_vec[2560 + 5];
    bool keep[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        float sg = sgv[q] + __shfl_xor(sgv[q], 32);
        float s1 = s1v[q] + __shfl_xor(s1v[q], 32);
        sg += bd;
        s1 += fabsf(bd);
        const bool mine = valid[q] && half == 0;
        const bool empty = in_range && sg < -(margin * s1 + margin);
        if (mine) {
            if (empty) sigma[pt[q]] = sg;
            if (dbg_sigma) { dbg_sigma[pt[q]] = in_range ? sg : dsn_nan_flag(); dbg_s1[pt[q]] = in_range ? s1 : dsn_nan_flag(); }
        }
        bool audit = false;
        if (audit_list && mine && empty) {
            uint32_t hsh = (uint32_t)pt[q] * 2654435761u;
            hsh ^= hsh >> 15;
            if ((hsh & 127u) == 5u) {
                const int a = atomicAdd(audit_count, 1);
                if (a < audit_cap) { audit_list[a] = (int32_t)pt[q]; audit = true; }
            }
        }
        keep[q] = mine && (!empty || audit);
    }
    const unsigned long long bm0 = __ballot(keep[0]), bm1 = __ballot(keep[1]);
    if (lane == 0) s_cnt[wave] = __popcll(bm0) + __popcll(bm1);
    __syncthreads();
    if (more) w16s_stage<4>(w, 0);      // (every wave has read its last weight block: the ring is free for the next tile)
    if (tid == 0) {
        const int tot = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        s_base = tot ? atomicAdd(keep_count, tot) : 0;
    }
    __syncthreads();
    {
        int off = s_base;
        for (int k = 0; k < wave; ++k) off += s_cnt[k];
        if (keep[0]) keep_list[off + __popcll(bm0 & ((1ull << lane) - 1ull))] = (int32_t)pt[0];
        if (keep[1]) keep_list[off + __popcll(bm0) + __popcll(bm1 & ((1ull << lane) - 1ull))] = (int32_t)pt[1];
    }
  }
}

void dsn_launch_screen16(const float* packed, const DsnFrameState* fs, const float* x_c, int64_t N, const int32_t* active_list,
                         const int32_t* active_count, float* sigma, int32_t* keep_list, int32_t* keep_count, float* dbg_sigma,
                         float* dbg_s1, hipStream_t st, int32_t* audit_list, int32_t* audit_count, int audit_cap) {
    if (N == 0) return;
    // DSN_SCREEN_WAVES (experiments): 4 = four one-tile waves, two workgroups per CU; 2 = k_screen16x2 (four two-tile waves: half the
    // LDS operand reads and DMA instructions per sample, bit-identical results - and 1.7 % SLOWER, 4.86 vs 4.78 ms in one call,
    // profiles/r02_screen_x2_ab.txt: what the second wave per SIMD hides is worth more than the operand bytes); default = eight
    // one-tile waves, one persistent workgroup per CU
    const char* wenv = getenv("DSN_SCREEN_WAVES");      // (read per launch: tests switch it)
    const int waves = wenv ? atoi(wenv) : 8;
    // experiment switch: overrides the margin the packed parameters carry (default / calibrated, see k_screen16)
    static const float margin = getenv("DSN_SCREEN_MARGIN") ? (float)atof(getenv("DSN_SCREEN_MARGIN")) : 0.0f;
    if (waves == 4)
        hipLaunchKernelGGL(k_screen16<4>, dim3((unsigned)std::min<int64_t>((N + 127) / 128, 2 * dsn_cu_count())), dim3(256), 0, st, packed, fs, x_c, N, active_list,
                           active_count, sigma, keep_list, keep_count, dbg_sigma, dbg_s1, margin, audit_list, audit_count, audit_cap);
    else if (waves == 2)
        hipLaunchKernelGGL(k_screen16x2, dim3((unsigned)std::min<int64_t>((N + 255) / 256, dsn_cu_count())), dim3(256), 0, st, packed, fs, x_c, N, active_list,
                           active_count, sigma, keep_list, keep_count, dbg_sigma, dbg_s1, margin, audit_list, audit_count, audit_cap);
    else
        hipLaunchKernelGGL(k_screen16<8>, dim3((unsigned)std::min<int64_t>((N + 255) / 256, dsn_cu_count())), dim3(512), 0, st, packed, fs, x_c, N, active_list,
                           active_count, sigma, keep_list, keep_count, dbg_sigma, dbg_s1, margin, audit_list, audit_count, audit_cap);
}

// DSN_SCREEN_AUDIT: out[0] += number of audited samples (declared empty by the screen, evaluated by the accurate pass anyway)
// whose accurate density is > 0; out[1] = max over them of that density (float bits; positive floats order like their bits);
// out[2] = the number of samples audited (the list's capacity at most)
__global__ void __launch_bounds__(256) k_screen_audit(const int32_t* __restrict__ audit_list, const int32_t* __restrict__ audit_count,
                                                      int audit_cap, const float* __restrict__ sigma, int32_t* __restrict__ out) {
    const int n = min(*audit_count, audit_cap);      // (the counter runs on past the capacity: only the first audit_cap were remembered)
    if (blockIdx.x == 0 && threadIdx.x == 0) out[2] = n;      // what was actually audited, for the host mirror
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float s = sigma[audit_list[i]];
        if (s > 0.0f) { atomicAdd(out, 1); atomicMax(out + 1, __float_as_int(s)); }
    }
}
void dsn_launch_screen_audit(const int32_t* audit_list, const int32_t* audit_count, int audit_cap, const float* sigma, int32_t* out,
                             hipStream_t st) {
    hipLaunchKernelGGL(k_screen_audit, dim3(64), dim3(256), 0, st, audit_list, audit_count, audit_cap, sigma, out);
}

// ---------------------------------------------------------------------------------------------
// dsn_calibrate_screen: the screen's margin for THESE parameters.  n points around the canonical surface (face centroid +
// a hash offset within +-0.15 m: where the canonical points of non-transparent samples lie, |h| <= 0.1), the screen's
// sigma~ and S1 and the exact-fp32 density for each; d = max |sigma~ - sigma| / (S1 + 1) over the finite ones is the
// largest margin any of them would have needed (a sample is wrongly dropped iff sigma > 0 and sigma~ < -m (S1 + 1)).
// margin = 10 d, at least F16_SCREEN_FLOOR; above F16_SCREEN_CAP the screen is useless for this network: margin = +inf.
// ---------------------------------------------------------------------------------------------
#define F16_SCREEN_FLOOR 0.002f
#define F16_SCREEN_CAP 0.15f
#define F16_SCREEN_HEADROOM 10.0f
// The margin rule (round 3).  With sigma~ the screen's density, sigma the accurate one, S1 the magnitude of the summed terms and
// dev = |sigma~ - sigma| / (S1 + 1), rel = |sigma| / (S1 + 1): a sample is dropped WRONGLY iff sigma > 0 and sigma~ < -m (S1 + 1), which
// needs dev > m + rel.  The margin gives every calibration point the same headroom K = 10 in DEVIATION against that condition,
//     K dev <= m + rel   for all points   <=>   m = K max(dev - rel / K),
// floored at 0.002 and switched off (+inf) above 0.15.  Round 2 used m = K max(dev), the same rule with rel taken as 0: equal for
// networks whose density is small against its terms (the hash-initialised sets: rel ~ 1 / 20), tighter than needed where the large
// deviations sit at large |sigma|.  The CONVERGED set (w4) is what the cap is for: its empty space is strongly negative (sigma ~
// -S1 / 2: a margin of 0.2 would still drop 99.8 % of it), but around sigma = 0 its fp16 evaluation is off by 2-5 % of S1 - the
// maximum over a million points of a heavy-tailed quantity (p99: 0.3 %), which came out at 0.022 on one frame and 0.050 on another.
// A margin of 0.2-0.5 on that footing is a statistical bet, not a bound: the screen stays off for such parameters
// (profiles/r03_w4_screen_calibration.txt).
__global__ void __launch_bounds__(256) k_calib_points(const float4* __restrict__ cent, int F, int64_t n, float* __restrict__ x, float box) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 c = cent[(int)(i % F)];
    uint32_t h = (uint32_t)i * 2654435761u + 0x9e3779b9u;
    float o[3];
    for (int k = 0; k < 3; ++k) {
        h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
        o[k] = ((float)(h >> 8) * (1.0f / 16777216.0f) - 0.5f) * box;
    }
    x[3 * i] = c.x + o[0]; x[3 * i + 1] = c.y + o[1]; x[3 * i + 2] = c.z + o[2];
}
// calibration points from a FRAME (round 3): the canonical points of its non-transparent samples (x_c [N,3], list / *count as the
// geometry phase of dsn_render_rays leaves them), every point taken as it is for the first half of the set and moved by a hash
// offset of up to +-halo per axis for the second half (the neighbourhood other rays / poses of the sequence will visit); a frame
// without non-transparent samples falls back to the centroid cube
#define DSN_CALIB_MIN_FRAME_LIST 65536      // (dsnerf.h: fewer non-transparent samples than this say nothing about the frame)
__global__ void __launch_bounds__(256) k_calib_points_frame(const float* __restrict__ x_c, const int32_t* __restrict__ list,
                                                             const int32_t* __restrict__ count, const float4* __restrict__ cent, int F,
                                                             int64_t n, float halo, float box, float* __restrict__ x) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t cnt = (int64_t)(*count);
    uint32_t h = (uint32_t)i * 2654435761u + 0x9e3779b9u;
    float o[3];
    for (int k = 0; k < 3; ++k) {
        h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
        o[k] = (float)(h >> 8) * (1.0f / 16777216.0f) - 0.5f;
    }
    if (cnt < DSN_CALIB_MIN_FRAME_LIST) {
        const float4 c = cent[(int)(i % F)];
        x[3 * i] = c.x + o[0] * box; x[3 * i + 1] = c.y + o[1] * box; x[3 * i + 2] = c.z + o[2] * box;
        return;
    }
    const int64_t half = n / 2 > 0 ? n / 2 : 1;
    const int64_t k = i % half;                                   // both halves walk the same evenly spread subset of the list
    const int64_t slot = (int64_t)(((__int128)k * cnt) / half) % cnt;
    const int64_t s = (int64_t)list[slot];
    const float j = i >= half ? 2.0f * halo : 0.0f;
    x[3 * i] = x_c[3 * s] + o[0] * j; x[3 * i + 1] = x_c[3 * s + 1] + o[1] * j; x[3 * i + 2] = x_c[3 * s + 2] + o[2] * j;
}
__global__ void __launch_bounds__(256) k_calib_reduce(const float* __restrict__ sg, const float* __restrict__ s1,
                                                      const float* __restrict__ sig, int64_t n, uint32_t* __restrict__ acc) {
    float d = 0.0f, t = 0.0f;
    int bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float a = sg[i], b = s1[i], c = sig[i];
        const float q = fabsf(a - c) / (b + 1.0f);
        if (q == q && fabsf(q) < INFINITY) {
            d = fmaxf(d, q);                                                 // largest deviation (reported)
            t = fmaxf(t, q - fabsf(c) / (F16_SCREEN_HEADROOM * (b + 1.0f)));  // what the margin is made from (see above; >= 0)
        } else ++bad;     // fp16 overflow inside the screen: sample is kept anyway
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { d = fmaxf(d, __shfl_xor(d, o)); t = fmaxf(t, __shfl_xor(t, o)); bad += __shfl_xor(bad, o); }
    if ((threadIdx.x & 63) == 0) { atomicMax(acc, __float_as_uint(d)); atomicAdd(acc + 1, (uint32_t)bad); atomicMax(acc + 3, __float_as_uint(t)); }
}
__global__ void k_calib_finish(const uint32_t* __restrict__ acc, float* __restrict__ packed_margin, float* __restrict__ out4, float n) {
    // out4[7] (input, >= 0): the statistic carried over from calibrations of other frame states of the same parameters - the margin and
    // the dropped share (out4[4]) are then those of the joint set
    const float d = __uint_as_float(acc[0]);
    float t = __uint_as_float(acc[3]);
    if (out4 && out4[7] > t) t = out4[7];
    float m = fmaxf(F16_SCREEN_HEADROOM * t, F16_SCREEN_FLOOR);
    if (!(m <= F16_SCREEN_CAP)) m = INFINITY;
    *packed_margin = m;
    if (out4) { out4[0] = d; out4[1] = m; out4[2] = (float)acc[1] / n; out4[3] = n; out4[5] = t; }
}
// how many of the calibration points the screen drops with the margin just set (out4[4] = fraction): tells the caller whether
// the screen pays for itself on this network (a network that is dense everywhere near the surface keeps every sample)
__global__ void __launch_bounds__(256) k_calib_dropped(const float* __restrict__ sg, const float* __restrict__ s1, int64_t n,
                                                       const float* __restrict__ packed_margin, uint32_t* __restrict__ acc) {
    const float m = *packed_margin;
    int c = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) c += sg[i] < -(m * s1[i] + m) ? 1 : 0;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(acc + 2, (uint32_t)c);
}
__global__ void k_calib_dropped_finish(const uint32_t* __restrict__ acc, float* __restrict__ out, float n) { out[4] = (float)acc[2] / n; }
size_t dsn_calibrate_workspace_size(int64_t n) { return 256 + dsn_align256(12 * (size_t)n) + 4 * dsn_align256(4 * (size_t)n); }
void dsn_launch_calibrate_screen(const DsnSceneView& s, float* packed, int64_t n, void* workspace, float* out4, hipStream_t st,
                                 const float* frame_x_c, const int32_t* frame_list, const int32_t* frame_count) {
    char* p = (char*)workspace;
    uint32_t* acc = (uint32_t*)p;             p += 256;
    float* x = (float*)p;                     p += dsn_align256(12 * (size_t)n);
    float* sg = (float*)p;                    p += dsn_align256(4 * (size_t)n);
    float* s1 = (float*)p;                    p += dsn_align256(4 * (size_t)n);
    float* sig = (float*)p;                   p += dsn_align256(4 * (size_t)n);
    int32_t* lst = (int32_t*)p;
    (void)hipMemsetAsync(acc, 0, 256, st);
    const char* be = getenv("DSN_CALIB_BOX");      // (experiments: edge of the cube of offsets around the canonical centroids, metres)
    const float box = be ? (float)atof(be) : 0.3f;
    if (frame_x_c)
        hipLaunchKernelGGL(k_calib_points_frame, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, frame_x_c, frame_list, frame_count,
                           s.cent_canon, s.F, n, 0.02f, box, x);
    else
        hipLaunchKernelGGL(k_calib_points, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, s.cent_canon, s.F, n, x, box);
    dsn_launch_screen16(packed, s.frame, x, n, nullptr, nullptr, sg, lst, (int32_t*)(acc + 8), sg, s1, st, nullptr, nullptr, 0);
    dsn_launch_field(packed, s.frame, x, n, nullptr, nullptr, sig, nullptr, nullptr, st);
    hipLaunchKernelGGL(k_calib_reduce, dim3(1024), dim3(256), 0, st, sg, s1, sig, n, acc);
    hipLaunchKernelGGL(k_calib_finish, dim3(1), dim3(1), 0, st, acc, packed + OFF_SCAL + 5, out4, (float)n);
    if (out4) {
        hipLaunchKernelGGL(k_calib_dropped, dim3(1024), dim3(256), 0, st, sg, s1, n, packed + OFF_SCAL + 5, acc);
        hipLaunchKernelGGL(k_calib_dropped_finish, dim3(1), dim3(1), 0, st, acc, out4, (float)n);
    }
}
__global__ void k_set_scalar(float* p, float v) { *p = v; }
void dsn_launch_set_screen_margin(float* packed, float margin, hipStream_t st) {
    hipLaunchKernelGGL(k_set_scalar, dim3(1), dim3(1), 0, st, packed + OFF_SCAL + 5, margin);
}
void dsn_launch_set_packed_scalar(float* packed, int word, float v, hipStream_t st) {
    hipLaunchKernelGGL(k_set_scalar, dim3(1), dim3(1), 0, st, packed + OFF_SCAL + word, v);
}

// ---------------------------------------------------------------------------------------------
// k_light16 : model/spacenet.py:254-265 + :174-188 LightingMLP with the same split-fp16 products.
// 9 -> 128 -> 128 -> 1 per point; the 20 weight blocks (80 KB) are L1/L2-resident, so every wave reads its
// operands straight from memory (no LDS ring: the matrix work per point is 50x smaller than the trunk's).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void light_block0(const char* blkp, int lane, const half8 (&xh)[2], const half8 (&xl)[2], f32x16& aM, f32x16& aC) {
    const half8 h0 = *reinterpret_cast<const half8*>(blkp + lane * 16);
    const half8 l0 = *reinterpret_cast<const half8*>(blkp + 1024 + lane * 16);
    aM = MFMA16(h0, xh[0], aM);
    aC = MFMA16(h0, xl[0], aC);
    aC = MFMA16(l0, xh[0], aC);
}
__device__ __forceinline__ void light_block(const char* __restrict__ blkp, int lane, const half8 (&xh)[2], const half8 (&xl)[2],
                                            f32x16& aM, f32x16& aC, bool second_step) {
    const half8 h0 = *reinterpret_cast<const half8*>(blkp + lane * 16);
    const half8 l0 = *reinterpret_cast<const half8*>(blkp + 1024 + lane * 16);
    aM = MFMA16(h0, xh[0], aM);
    aC = MFMA16(h0, xl[0], aC);
    aC = MFMA16(l0, xh[0], aC);
    if (second_step) {
        const half8 h1 = *reinterpret_cast<const half8*>(blkp + 2048 + lane * 16);
        const half8 l1 = *reinterpret_cast<const half8*>(blkp + 3072 + lane * 16);
        aM = MFMA16(h1, xh[1], aM);
        aC = MFMA16(h1, xl[1], aC);
        aC = MFMA16(l1, xh[1], aC);
    }
}

// Round 3: persistent workgroups with the weights in LDS.  Every wave used to read its 72 KB of weight operands straight from
// memory for every 32 samples - more than a CU's L1 holds, so 2.25 KB per sample came from L2: 4.4 GB per launch on the bench frame,
// 10 TB/s, the kernel was L2-bound (0.41 ms).  Now a workgroup stages the 20 blocks once (72 KB: two workgroups per CU) and walks
// its tiles of 128 samples with ds_read_b128 operands.  Same products in the same order: bit-identical colours.
#define LIGHT_LDS_BYTES (4 * 2048 + 16 * 4096)
__global__ void __launch_bounds__(256, 1)
k_light16(const float* __restrict__ packed, const DsnFrameState* __restrict__ fs, const float* __restrict__ n_w,
          const float* __restrict__ x_w_pts, const float* __restrict__ ray_o, const float* __restrict__ ray_d,
          const float* __restrict__ z_vals, const float* essence, int64_t N, int S,
          const int32_t* __restrict__ active_list, const int32_t* __restrict__ active_count,
          float* colour, float* __restrict__ tr_hl1, float* __restrict__ tr_hl2, float* __restrict__ tr_pre) {
    // (essence and colour may be the SAME array - the fused path's workspace keeps the colour where the essence was: a tile reads its
    //  samples' essences at its top and writes their colours at its end; hence no __restrict__ on the two)
    // tr_*: (training forward) the two hidden layers after their ReLU, row-major [N,128], and the pre-activation of the output
    // [N] - what the backward of the lighting MLP needs, so that it does not have to evaluate the MLP again
    __shared__ __attribute__((aligned(16))) char s_w[LIGHT_LDS_BYTES];      // [LT0: 4 x (hi, lo of k-step 0) | LT1: 16 x 4 KB]
#ifndef F16_SHARE_SIMD
    DSN_OWN_SIMD();      // (round 5: this kernel too made co-resident waves of other kernels read registers early - dsn_common.h)
#endif
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5;
    const int64_t count = active_list ? (int64_t)(*active_count) : N;
    const int64_t ntiles = (count + 127) / 128;
    if ((int64_t)blockIdx.x >= ntiles) return;      // block-uniform
    {
        const char* g16 = reinterpret_cast<const char*>(packed + OFF16_BASE);
        const uint4* g0 = reinterpret_cast<const uint4*>(g16 + (size_t)(OFF_LT0 / DSN_BLK) * 4096);
        const uint4* g1 = reinterpret_cast<const uint4*>(g16 + (size_t)(OFF_LT1 / DSN_BLK) * 4096);
        uint4* d = reinterpret_cast<uint4*>(s_w);
        for (int i = threadIdx.x; i < 4 * 128; i += 256) d[i] = g0[(i >> 7) * 256 + (i & 127)];      // first 2 KB of each 4 KB block
        for (int i = threadIdx.x; i < 16 * 256; i += 256) d[4 * 128 + i] = g1[i];
    }
    __syncthreads();
    // The inputs of a tile (its samples' normals, depths, rays) are gathers behind a list entry: two dependent trips to memory in
    // front of 108 MFMAs, and a CU holds only two waves per SIMD of this kernel - the kernel was bound by that latency (0.30 ms
    // for 0.05 ms of matrix work).  They are fetched one tile ahead now (the list entry two tiles ahead).
    struct LightIn { float nw[3], o[3], d[3], z; };
    auto tile_pt = [&](int64_t t, bool& ok) -> int64_t {
        int64_t sl = (t * 4 + wave) * 32 + (lane & 31);
        ok = t < ntiles && sl < count;
        if (!ok) sl = count - 1;
        return active_list ? (int64_t)active_list[sl] : sl;
    };
    auto fetch_in = [&](int64_t p, LightIn& a) {
        const int64_t r = p / S;
        a.nw[0] = n_w[3 * p]; a.nw[1] = n_w[3 * p + 1]; a.nw[2] = n_w[3 * p + 2];
        a.d[0] = ray_d[3 * r]; a.d[1] = ray_d[3 * r + 1]; a.d[2] = ray_d[3 * r + 2];
        if (x_w_pts) { a.o[0] = x_w_pts[3 * p]; a.o[1] = x_w_pts[3 * p + 1]; a.o[2] = x_w_pts[3 * p + 2]; a.z = 0.0f; }
        else { a.o[0] = ray_o[3 * r]; a.o[1] = ray_o[3 * r + 1]; a.o[2] = ray_o[3 * r + 2]; a.z = z_vals[p]; }
    };
    bool valid_c, valid_n, valid_nn;
    int64_t pt_c = tile_pt(blockIdx.x, valid_c);
    LightIn in_c, in_n;
    fetch_in(pt_c, in_c);
    int64_t pt_n = tile_pt((int64_t)blockIdx.x + gridDim.x, valid_n);
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t slot0 = (tile * 4 + wave) * 32;
    const bool valid = valid_c;
    const int64_t pt = pt_c;
    const LightIn cur = in_c;
    // next tile's inputs and the list entry of the one after it: in flight under this tile's arithmetic
    fetch_in(pt_n, in_n);
    const int64_t pt_nn = tile_pt(tile + 2 * (int64_t)gridDim.x, valid_nn);
    pt_c = pt_n; valid_c = valid_n; in_c = in_n;
    pt_n = pt_nn; valid_n = valid_nn;
    if (slot0 >= count) continue;                    // (no barrier inside the loop)
    // (needed at the very end: fetched now, waited for behind the arithmetic)
    const float ess[3] = {essence[3 * pt], essence[3 * pt + 1], essence[3 * pt + 2]};

    float in9[10];
    in9[0] = cur.nw[0]; in9[1] = cur.nw[1]; in9[2] = cur.nw[2];
    float xw[3];
    const float d[3] = {cur.d[0], cur.d[1], cur.d[2]};
    if (x_w_pts) { xw[0] = cur.o[0]; xw[1] = cur.o[1]; xw[2] = cur.o[2]; }
    else {
        const float z = cur.z;
        xw[0] = cur.o[0] + d[0] * z; xw[1] = cur.o[1] + d[1] * z; xw[2] = cur.o[2] + d[2] * z;
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

    // input operand: k-slot j of step 0 holds feature 2j + half (j < 5), zero beyond
    half8 xh[2], xl[2];
    {
        f32x16 v = zero16();
#pragma unroll
        for (int j = 0; j < 5; ++j) v[j] = half ? in9[2 * j + 1] : in9[2 * j];
        split16<true>(v, xh, xl);
    }
    half8 h1h[4][2], h1l[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        f32x16 aM = rows16(packed + OFF_BLT0, m, half), aC = zero16();
        light_block0(s_w + m * 2048, lane, xh, xl, aM, aC);
        f32x16 v = fold16(aM, aC);
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.0f);
        if (tr_hl1 && valid) store16(tr_hl1 + pt * 128 + 4 * half + 32 * m, v, 1.0f);
        split16<true>(v, h1h[m], h1l[m]);
    }
    float part = 0.0f;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        f32x16 aM = rows16(packed + OFF_BLT1, m, half), aC = zero16();
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
            light_block(s_w + 4 * 2048 + (m * 4 + kb) * 4096, lane, h1h[kb], h1l[kb], aM, aC, true);
        f32x16 v = fold16(aM, aC);
        const f32x16 w2 = rows16(packed + OFF_WLT2, m, half);
#pragma unroll
        for (int r = 0; r < 16; ++r) { v[r] = fmaxf(v[r], 0.0f); part = fmaf(w2[r], v[r], part); }
        if (tr_hl2 && valid) store16(tr_hl2 + pt * 128 + 4 * half + 32 * m, v, 1.0f);
    }
    part += __shfl_xor(part, 32);
    const float o = part + packed[OFF_SCAL + 4];
    const float wgt = (o > 0.0f ? o : expm1f(o)) + 1.0f;   // ELU(alpha=1) + 1
    if (tr_pre && valid && half == 0) tr_pre[pt] = o;
    if (valid && half == 0) {
        colour[3 * pt + 0] = wgt * ess[0];
        colour[3 * pt + 1] = wgt * ess[1];
        colour[3 * pt + 2] = wgt * ess[2];
    }
  }
}

void dsn_launch_light16(const float* packed, const DsnFrameState* fs, const float* n_w, const float* x_w,
                        const float* ray_o, const float* ray_d, const float* z_vals, const float* essence, int64_t N,
                        int S, const int32_t* active_list, const int32_t* active_count, float* colour, hipStream_t st,
                        float* tr_hl1, float* tr_hl2, float* tr_pre) {
    int64_t blocks = (N + 127) / 128;
    if (blocks == 0) return;
    // (one workgroup per compute unit since round 5: its waves own their SIMDs' register files, two no longer fit)
    hipLaunchKernelGGL(k_light16, dim3((unsigned)std::min<int64_t>(blocks, (int64_t)dsn_cu_count())), dim3(256), 0, st, packed, fs, n_w,
                       x_w, ray_o, ray_d, z_vals, essence, N, S, active_list, active_count, colour, tr_hl1, tr_hl2, tr_pre);
}
