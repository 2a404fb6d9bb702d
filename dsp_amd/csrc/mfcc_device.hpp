// mfcc_device.hpp -- device helpers shared by the MFCC kernels (complex math, radix-4
// butterfly, wave-scope LDS ordering, DPP / permlane moves, frame cursor).
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

#include "mfcc_kernels.hpp"

namespace dsp {

namespace {

struct c32 { float x, y; };
typedef float f2v __attribute__((ext_vector_type(2)));
typedef int i2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ c32 cadd(c32 a, c32 b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ c32 csub(c32 a, c32 b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ c32 cmul(c32 a, c32 w) { return {a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x}; }
// multiply by -i
__device__ __forceinline__ c32 cmul_mi(c32 a) { return {a.y, -a.x}; }

// forward radix-4 butterfly, W4 = -i
__device__ __forceinline__ void radix4(c32 (&s)[4])
{
    const c32 t0 = cadd(s[0], s[2]), t1 = csub(s[0], s[2]);
    const c32 t2 = cadd(s[1], s[3]), t3 = cmul_mi(csub(s[1], s[3]));
    s[0] = cadd(t0, t2);
    s[1] = cadd(t1, t3);
    s[2] = csub(t0, t2);
    s[3] = csub(t1, t3);
}

// Orders this wave's LDS traffic for the compiler.  The hardware executes one
// wave's DS instructions in order, so no s_barrier / waitcnt is needed between a
// ds_write and a ds_read of another lane's data inside the same wave.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ void swap_hi32(float &a, float &b)
{   // a[lanes 32..63] <-> b[lanes 0..31]
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void swap_odd16(float &a, float &b)
{   // a[odd 16-lane rows] <-> b[even 16-lane rows]
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}

// 2x2 block transposes between a register pair (a: slot bit 0, b: slot bit 1) and
// one lane bit: afterwards a[bit=1] holds the partner lane's old b, b[bit=0] the
// partner lane's old a.
template <int CTRL, int BANKS>
__device__ __forceinline__ float dpp_into(float old, float src);
template <int CTRL>
__device__ __forceinline__ float dpp(float v);
__device__ __forceinline__ void swap_lane8(float &a, float &b);
__device__ __forceinline__ void swap_lane4(float &a, float &b);

template <int CTRL>
__device__ __forceinline__ float dpp(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// masked DPP move: lanes whose bank (lane%16/4) is in BANKS take src[perm], others keep old
template <int CTRL, int BANKS>
__device__ __forceinline__ float dpp_into(float old, float src)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, 0xF, BANKS, false));
}
constexpr int DPP_ROW_SHL4 = 0x104, DPP_ROW_SHR4 = 0x114, DPP_ROW_ROR8 = 0x128;
constexpr int DPP_QUAD_1032 = 0xB1;   // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_2301 = 0x4E;   // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_HALF_MIRROR = 0x141;
constexpr int DPP_ROW_MIRROR = 0x140;

__device__ __forceinline__ void swap_lane8(float &a, float &b)
{   // lane ^ 8 = rotate the 16-lane row by 8; banks 0,1 have bit3 = 0, banks 2,3 bit3 = 1
    const float nb = dpp_into<DPP_ROW_ROR8, 0x3>(b, a);
    a = dpp_into<DPP_ROW_ROR8, 0xC>(a, b);
    b = nb;
}
__device__ __forceinline__ void swap_lane4(float &a, float &b)
{   // lane ^ 4: banks 0,2 (bit2 = 0) read lane+4, banks 1,3 read lane-4
    const float nb = dpp_into<DPP_ROW_SHL4, 0x5>(b, a);
    a = dpp_into<DPP_ROW_SHR4, 0xA>(a, b);
    b = nb;
}
// lane bits 1 and 0 have no DPP write mask: quad permute + select on the lane bit
template <int CTRL>
__device__ __forceinline__ void swap_quad(float &a, float &b, bool bit_set)
{
    const float pa = dpp<CTRL>(a), pb = dpp<CTRL>(b);
    b = bit_set ? b : pa;
    a = bit_set ? pb : a;
}

// max over the wave of NON-NEGATIVE floats: their bit patterns order like
// unsigned integers, so the reduction runs on v_max_u32 (fuses with DPP, needs no
// NaN canonicalisation moves).
template <int CTRL>
__device__ __forceinline__ unsigned dppu(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
__device__ __forceinline__ float wave_max_nonneg(float f)
{
    unsigned v = __float_as_uint(f);
    v = max(v, dppu<DPP_QUAD_1032>(v));
    v = max(v, dppu<DPP_QUAD_2301>(v));
    v = max(v, dppu<DPP_ROW_HALF_MIRROR>(v));
    v = max(v, dppu<DPP_ROW_MIRROR>(v));            // every lane: max of its 16-lane row
    const unsigned r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16);
    const unsigned r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
    return __uint_as_float(max(max(r0, r1), max(r2, r3)));
}

// per-wave LDS carve (bytes)
constexpr int LDS_XCHG = 0;                 // 256 x float2 exchange tile, later P[0..256]
constexpr int LDS_PART = 2048 + 64;         // 65 partial sums (+ zero slot)
constexpr int LDS_LOGMEL = LDS_PART + 320;  // up to 80 log-mel values
constexpr int LDS_WAVE_BYTES = LDS_LOGMEL + 320;
constexpr int LDS_TILE_BYTES = 64 * 16 * 4;   // mel energies of 16 frames, E[mel][frame ^ (mel>>2)]
constexpr int LDS_POOL_BYTES = 32 * 16;       // POOL 1: (sum, sum of squares) in double for up to 32 coefficients
constexpr int LDS_STOP_BYTES = 64 * kStopFusedUnits * 8;    // POOL 2: float64 layer-1 partial sums per lane (stop-word net)
static_assert(LDS_WAVE_BYTES % 16 == 0, "keep the carve 16-byte aligned");
// wave-uniform cursor over the frames this wave owns: chunks of `chunk`
// consecutive frames dealt round-robin to the waves of the grid, so one wave's
// 52-byte outputs land in consecutive cache lines.
struct FrameCursor {
    long f, chunk_end, chunk_first, stride, n;
    long clip;              // clip mode: f = clip * fpc + t
    long jump_clips;        // stride = jump_clips * fpc + jump_t
    int t, t0, fpc, chunk, jump_t;
    long clip0;             // (clip0, t0): position of chunk_first
    __device__ __forceinline__ void init(long wave, long n_waves, int chunk_, long n_, int fpc_)
    {
        chunk = chunk_; n = n_; fpc = fpc_;
        stride = n_waves * chunk;
        chunk_first = wave * chunk;
        f = chunk_first;
        chunk_end = f + chunk < n ? f + chunk : n;
        clip0 = 0; t0 = 0; jump_clips = 0; jump_t = 0;
        if (fpc > 0) {          // the only divisions: once per wave, not per frame
            clip0 = f / fpc; t0 = (int)(f - clip0 * fpc);
            jump_clips = stride / fpc; jump_t = (int)(stride - jump_clips * fpc);
        }
        clip = clip0; t = t0;
    }
    __device__ __forceinline__ bool valid() const { return f < n; }
    // advance by `step` frames (step divides chunk)
    __device__ __forceinline__ void next(int step)
    {
        f += step;
        if (f >= chunk_end) {
            chunk_first += stride;
            f = chunk_first;
            chunk_end = f + chunk < n ? f + chunk : n;
            clip0 += jump_clips; t0 += jump_t;
            if (t0 >= fpc) { t0 -= fpc; ++clip0; }
            clip = clip0; t = t0;
        } else if (fpc > 0) {
            t += step;
            while (t >= fpc) { t -= fpc; ++clip; }
        }
    }
};

// Cursor of the wave-per-frame kernel: same walk as FrameCursor (chunks of `chunk` consecutive
// frames dealt round-robin to the waves), one frame per step, and it carries the sample offset of
// the frame incrementally, so a step is a handful of scalar adds and selects: no multiply, no
// loop, no branch.  CLIPS: frames overlap inside clips (offset = clip * clip_stride + t * hop);
// otherwise frames lie back to back (offset = f * frame_len).
template <bool CLIPS>
struct WaveCursor {
    long f, off;                 // frame index and its first sample; meaningful while valid()
    long clip;                   // CLIPS: clip of frame f
    long chunk_f, chunk_off;     // first frame of the current chunk; CLIPS: chunk_off = offset of its clip
    long chunk_clip;
    long stride_f, stride_off;   // chunk-to-chunk jump of this wave; CLIPS: stride_off = jump_clips * clip_stride
    long clip_off, clip_stride;  // CLIPS: offset of the current clip
    long jump_clips;
    int remaining;               // frames this wave still has to visit, f included (32-bit: scalar compare)
    int left, chunk;             // frames of the chunk still to come after f
    int t, t0, fpc, jump_t, hop; // CLIPS: frame in clip, of f / of the chunk start; hop = samples between frames
    __device__ __forceinline__ void init(long wave, long n_waves, int chunk_, long n, int fpc_, int hop_, long clip_stride_)
    {
        chunk = chunk_; fpc = fpc_; hop = hop_; clip_stride = clip_stride_;
        stride_f = n_waves * chunk;
        chunk_f = wave * chunk;
        f = chunk_f;
        left = chunk - 1;
        // frames of this wave: its chunks, minus what the batch's last (partial) chunk lacks
        const long n_chunks = (n + chunk - 1) / chunk;
        const long mine = wave < n_chunks ? (n_chunks - 1 - wave) / n_waves + 1 : 0;
        const bool owns_last = mine > 0 && (n_chunks - 1 - wave) % n_waves == 0;
        remaining = (int)(mine * chunk - (owns_last ? n_chunks * chunk - n : 0));
        clip = chunk_clip = 0; t = t0 = 0; jump_t = 0; jump_clips = 0; clip_off = 0;
        if (CLIPS) {            // the only divisions and wide multiplies: once per wave
            chunk_clip = f / fpc; t0 = (int)(f - chunk_clip * fpc);
            jump_clips = stride_f / fpc; jump_t = (int)(stride_f - jump_clips * fpc);
            stride_off = jump_clips * clip_stride;
            chunk_off = chunk_clip * clip_stride;
            clip = chunk_clip; t = t0; clip_off = chunk_off;
            off = clip_off + (long)t * hop;
        } else {
            stride_off = stride_f * hop;
            chunk_off = chunk_f * hop;
            off = chunk_off;
        }
    }
    __device__ __forceinline__ bool valid() const { return remaining > 0; }
    // all selects on wave-uniform values (s_cselect), no branch
    __device__ __forceinline__ void next()
    {
        --remaining;
        const bool in_chunk = left > 0;
        left = in_chunk ? left - 1 : chunk - 1;
        chunk_f += in_chunk ? 0 : stride_f;
        f = in_chunk ? f + 1 : chunk_f;
        if (CLIPS) {
            // inside the chunk: next frame of the clip, or frame 0 of the next clip
            const bool wrap = t + 1 == fpc;
            const int t_in = wrap ? 0 : t + 1;
            const long clip_in = clip + (wrap ? 1 : 0);
            const long clip_off_in = clip_off + (wrap ? clip_stride : 0);
            const long off_in = wrap ? clip_off_in : off + hop;
            // chunk jump: advance the chunk start by (jump_clips, jump_t) with carry
            const int t0_raw = t0 + (in_chunk ? 0 : jump_t);
            const bool carry = t0_raw >= fpc;
            t0 = carry ? t0_raw - fpc : t0_raw;
            chunk_clip += in_chunk ? 0 : jump_clips + (carry ? 1 : 0);
            chunk_off += in_chunk ? 0 : stride_off + (carry ? clip_stride : 0);
            t = in_chunk ? t_in : t0;
            clip = in_chunk ? clip_in : chunk_clip;
            clip_off = in_chunk ? clip_off_in : chunk_off;
            off = in_chunk ? off_in : chunk_off + t0 * hop;      // t0 * hop < samples per clip < 2^31
        } else {
            chunk_off += in_chunk ? 0 : stride_off;
            off = in_chunk ? off + hop : chunk_off;
        }
    }
};

// Cursor of the fused clip kernels (one wavefront walks one clip from its first frame to its last; clips dealt round-robin to the
// waves).  Uniform batches: clip c starts at c * clip_stride and has fpc frames; ragged batches (spans != nullptr): start, samples and
// frame count come from the clip's ClipSpan, one scalar load per clip -- the HOST has put the spans in an order that makes the fixed
// deal even (by length, snaking over the waves; capi.cpp ragged_spans) and the results go to ClipSpan::orig.  (Handing the clips out
// by an atomic counter inside the kernel was tried first: the extra live state cost the frame loop 4 - 15 spilled VGPRs.)
// Same members as WaveCursor where the kernels read them.
struct ClipCursor {
    long f, off, clip;
    long n_clips, n_waves, clip_stride;
    const ClipSpan *spans;
    int left, remaining, t, hop, fpc, n_samples;
    __device__ __forceinline__ void enter()
    {
        t = 0;
        if (clip >= n_clips) { remaining = 0; return; }
        remaining = 2;                              // "not the wave's last frame": a clip's end is left == 0
        if (spans) {
            const ClipSpan s = spans[clip];
            off = s.off; n_samples = s.n; left = s.frames - 1;
        } else {
            off = clip * clip_stride; left = fpc - 1;
        }
    }
    __device__ __forceinline__ void init(long wave, long n_waves_, long n_clips_, int fpc_, int hop_, long clip_stride_, const ClipSpan *spans_,
                                         int samples_per_clip)
    {
        n_clips = n_clips_; n_waves = n_waves_; clip_stride = clip_stride_; spans = spans_; hop = hop_; fpc = fpc_; n_samples = samples_per_clip;
        f = 0; off = 0; left = 0; clip = wave;
        enter();
    }
    __device__ __forceinline__ bool valid() const { return remaining > 0; }
    __device__ __forceinline__ void next()
    {
        ++f;
        if (left > 0) { --left; ++t; off += hop; }
        else { clip += n_waves; enter(); }
    }
};

// the kernel's argument struct where the hardware put it (the kernarg segment; Mfcc512Args is the FIRST parameter of every MFCC kernel): a pointer the
// cold per-clip code reloads model fields through, instead of keeping them in SGPRs across the frame loop
__device__ __forceinline__ const Mfcc512Args *kernarg_of_mfcc512()
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (const Mfcc512Args *)__builtin_amdgcn_kernarg_segment_ptr();
#else
    return nullptr;      // host pass of the single-source compile: never called
#endif
}

}  // namespace

}  // namespace dsp
