// mfcc1024_kernel.hip -- general wave-per-frame MFCC kernel for n_fft = 1024 (BASELINE config 3:
// 1024-point FFT, up to 128 mel bins).  Same chain as mfcc_kernels.hip (reference
// 2fa/audio/word/c/mfcc.c:142-221 with the constants as parameters), built for generality
// first: the 512-point complex FFT is a radix-8 Stockham autosort through a per-wave LDS
// image (3 stages, 8 points per lane), twiddles from an LDS table; mel, log and DCT loop
// over table-driven chunk slots.  Not yet tuned like the 512-point kernel.
#include <hip/hip_runtime.h>

#include "mfcc_device.hpp"

namespace dsp {

namespace {

constexpr int G_ZBUF = 0;                         // 513 x float2 Z image (Z[512] = Z[0]); later P[513]
constexpr int G_PART = 513 * 8 + 8;               // 257 partial sums (256 + zero slot)
constexpr int G_LMEL = G_PART + 260 * 4;          // 128 log-mel values
constexpr int G_WAVE_BYTES = G_LMEL + 128 * 4;
static_assert(G_WAVE_BYTES % 16 == 0, "keep the carve 16-byte aligned");
// block-shared: W512 table (4 KiB), W1024 table (2 KiB)
constexpr int G_W512 = 4 * G_WAVE_BYTES;
constexpr int G_W1024 = G_W512 + 512 * 8;
constexpr int G_BLOCK_BYTES = G_W1024 + 256 * 8;

// forward radix-8 butterfly: u[q] = sum_a v[a] W8^(a q)
__device__ __forceinline__ void radix8(c32 (&v)[8])
{
    c32 e[4] = {v[0], v[2], v[4], v[6]}, o[4] = {v[1], v[3], v[5], v[7]};
    radix4(e);
    radix4(o);
    constexpr float R2 = 0.70710678118654752f;
    const c32 t1 = {(o[1].x + o[1].y) * R2, (o[1].y - o[1].x) * R2};        // W8^1 = (1 - i)/sqrt2
    const c32 t2 = {o[2].y, -o[2].x};                                        // W8^2 = -i
    const c32 t3 = {(o[3].y - o[3].x) * R2, -(o[3].x + o[3].y) * R2};       // W8^3 = (-1 - i)/sqrt2
    v[0] = cadd(e[0], o[0]); v[4] = csub(e[0], o[0]);
    v[1] = cadd(e[1], t1);   v[5] = csub(e[1], t1);
    v[2] = cadd(e[2], t2);   v[6] = csub(e[2], t2);
    v[3] = cadd(e[3], t3);   v[7] = csub(e[3], t3);
}

}  // namespace

template <bool FULL>
__global__ __launch_bounds__(256) void mfcc1024_kernel(const Mfcc512Args args, const GenTables1024 *__restrict__ G)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char *wl = smem + wib * G_WAVE_BYTES;
    float2 *zbuf = reinterpret_cast<float2 *>(wl + G_ZBUF);
    float *pbuf = reinterpret_cast<float *>(wl + G_ZBUF);
    float *part = reinterpret_cast<float *>(wl + G_PART);
    float *lmel = reinterpret_cast<float *>(wl + G_LMEL);
    float2 *w512 = reinterpret_cast<float2 *>(smem + G_W512);
    float2 *w1024 = reinterpret_cast<float2 *>(smem + G_W1024);
    for (int i = threadIdx.x; i < 512; i += 256) w512[i] = make_float2(G->w512[0][i], G->w512[1][i]);
    if (threadIdx.x < 256) w1024[threadIdx.x] = make_float2(G->w1024[0][threadIdx.x], G->w1024[1][threadIdx.x]);
    __syncthreads();

    float win[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) win[i] = G->win[i][lane];
    const int n_mels = args.n_mels, n_mfcc = args.n_mfcc;
    const int n_slots = G->n_chunk_slots;
    const int dct_c = lane >> 2;
    const bool dct_store = (lane & 3) == 0 && dct_c < n_mfcc;
    if (lane == 0) part[kGenZeroSlot] = 0.0f;
    lmel[lane] = 0.0f;
    lmel[64 + lane] = 0.0f;
    wave_lds_sync();

    const long wave = (long)blockIdx.x * 4 + wib;
    const long n_waves = (long)gridDim.x * 4;
    FrameCursor cur;
    cur.init(wave, n_waves, args.chunk, args.n_frames, args.frames_per_clip);
    if (!cur.valid()) return;
    const unsigned amin_u = __float_as_uint(args.amin);
    const float neg_top_db = -args.top_db;
    const int frame_len = args.frame_len;

    auto frame_src = [&](const FrameCursor &c) -> const float * {
        if (args.frames_per_clip <= 0) return static_cast<const float *>(args.in) + c.f * (long)frame_len;
        return static_cast<const float *>(args.in) + c.clip * args.clip_stride + (long)c.t * args.hop;
    };
    auto load_frame8 = [&](const float *src, c32 (&z)[8]) {
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            const int i = 2 * (lane + 64 * a);
            if (FULL || i + 1 < frame_len) {
                const f2v x = args.frames_per_clip > 0 ? *reinterpret_cast<const f2v *>(src + i)      // clips re-read samples: cacheable
                                                       : __builtin_nontemporal_load(reinterpret_cast<const f2v *>(src + i));
                z[a] = {x.x, x.y};
            } else if (i < frame_len) {
                z[a] = {src[i], 0.0f};
            } else {
                z[a] = {0.0f, 0.0f};
            }
        }
    };

    c32 nxt[8];
    load_frame8(frame_src(cur), nxt);
    while (true) {
        const long f = cur.f;
        c32 v[8];
#pragma unroll
        for (int a = 0; a < 8; ++a) v[a] = {nxt[a].x * win[2 * a], nxt[a].y * win[2 * a + 1]};
        cur.next(1);
        const bool more = cur.valid();
        if (more) load_frame8(frame_src(cur), nxt);

        // ---- 512-point complex FFT: radix-8 Stockham, 3 stages ---------------------------
        // stage s (Ns = 8^s): k = lane % Ns; v[a] *= W_{8 Ns}^(a k); radix-8; y[(lane-k)*8 + k + q Ns] = u[q]
        radix8(v);                                                        // Ns = 1: no twiddle
#pragma unroll
        for (int q = 0; q < 8; ++q) zbuf[lane * 8 + q] = make_float2(v[q].x, v[q].y);
        wave_lds_sync();
#pragma unroll
        for (int s = 1; s < 3; ++s) {
            const int Ns = s == 1 ? 8 : 64;
            const int k = lane & (Ns - 1);
#pragma unroll
            for (int a = 0; a < 8; ++a) {
                const float2 x = zbuf[lane + 64 * a];
                v[a] = {x.x, x.y};
            }
            wave_lds_sync();
#pragma unroll
            for (int a = 1; a < 8; ++a) {
                const float2 w = w512[a * k * (512 / (8 * Ns))];
                v[a] = cmul(v[a], c32{w.x, w.y});
            }
            radix8(v);
            if (s == 1) {
                const int j0 = (lane - k) * 8 + k;
#pragma unroll
                for (int q = 0; q < 8; ++q) zbuf[j0 + q * 8] = make_float2(v[q].x, v[q].y);
                wave_lds_sync();
            }
        }
        // after the last stage lane l holds Z[l + 64 q] / 2: natural-order image, Z[512] = Z[0]
#pragma unroll
        for (int q = 0; q < 8; ++q) zbuf[lane + 64 * q] = make_float2(v[q].x, v[q].y);
        if (lane == 0) zbuf[512] = make_float2(v[0].x, v[0].y);
        wave_lds_sync();

        // ---- untangle: bins k = l + 64 t (t < 4) with 512 - k; bin 256 alone ------------------
        float P[8];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k = lane + 64 * t;
            const float2 a = zbuf[k], b = zbuf[512 - k], w = w1024[k];
            const c32 E = {a.x + b.x, a.y - b.y};
            const c32 O = {a.x - b.x, a.y + b.y};
            const c32 Tw = cmul(O, c32{w.x, w.y});
            const float xr = E.x + Tw.y, xi = E.y - Tw.x;
            const float mr = E.x - Tw.y, mi = E.y + Tw.x;
            P[2 * t] = xr * xr + xi * xi;
            P[2 * t + 1] = mr * mr + mi * mi;
        }
        const float2 zm = zbuf[256];
        wave_lds_sync();
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            pbuf[lane + 64 * t] = P[2 * t];
            pbuf[512 - lane - 64 * t] = P[2 * t + 1];
        }
        if (lane == 0) pbuf[256] = 4.0f * (zm.x * zm.x + zm.y * zm.y);
        wave_lds_sync();

        // ---- sparse mel: up to 4 chunk slots per lane ---------------------------------------
        for (int c = 0; c < n_slots; ++c) {
            const float *rd = pbuf + G->mel_k0[c][lane];
            float acc = 0.0f;
#pragma unroll
            for (int i = 0; i < kMelChunk; ++i) acc = fmaf(G->mel_w[c][i][lane], rd[i], acc);
            part[c * 64 + lane] = acc;
        }
        wave_lds_sync();
        float e[kGenMelsPerLane];
        float emax = 0.0f;
#pragma unroll
        for (int i = 0; i < kGenMelsPerLane; ++i) {
            float s = 0.0f;
#pragma unroll
            for (int g = 0; g < kGenGather; ++g) s += part[G->mel_src[i][g][lane]];
            e[i] = (lane + 64 * i < n_mels) ? s : 0.0f;
            emax = fmaxf(emax, e[i]);
        }
        const float ref = __uint_as_float(max(__float_as_uint(wave_max_nonneg(emax)), amin_u));
        const float inv = __builtin_amdgcn_rcpf(ref);
#pragma unroll
        for (int i = 0; i < kGenMelsPerLane; ++i) {
            const float ec = __uint_as_float(max(__float_as_uint(e[i]), amin_u));
            float db = 3.01029995663981195f * __builtin_amdgcn_logf(ec * inv);
            db = __builtin_amdgcn_fmed3f(db, neg_top_db, 0.0f);
            if (lane + 64 * i < n_mels) lmel[lane + 64 * i] = db;
        }
        wave_lds_sync();

        // ---- DCT-II: lane 4c+q dots log-mels [32q, 32q+32) ----------------------------------------
        {
            const float4 *rd = reinterpret_cast<const float4 *>(lmel + (lane & 3) * kGenDctLen);
            float c = 0.0f;
#pragma unroll
            for (int i = 0; i < kGenDctLen / 4; ++i) {
                const float4 x = rd[i];
                c = fmaf(G->dct_w[4 * i][lane], x.x, c);
                c = fmaf(G->dct_w[4 * i + 1][lane], x.y, c);
                c = fmaf(G->dct_w[4 * i + 2][lane], x.z, c);
                c = fmaf(G->dct_w[4 * i + 3][lane], x.w, c);
            }
            c += dpp<DPP_QUAD_1032>(c);
            c += dpp<DPP_QUAD_2301>(c);
            if (dct_store) args.out[f * n_mfcc + dct_c] = c;
        }
        wave_lds_sync();
        if (!more) return;
    }
}

hipError_t launch_mfcc1024(const Mfcc512Args &args, const GenTables1024 *tables, int blocks, hipStream_t stream)
{
    if (args.frame_len == 1024) hipLaunchKernelGGL((mfcc1024_kernel<true>), dim3(blocks), dim3(256), G_BLOCK_BYTES, stream, args, tables);
    else hipLaunchKernelGGL((mfcc1024_kernel<false>), dim3(blocks), dim3(256), G_BLOCK_BYTES, stream, args, tables);
    return hipGetLastError();
}

int mfcc1024_blocks_per_cu(bool full)
{
    int n = 0;
    hipError_t e = full ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, mfcc1024_kernel<true>, 256, G_BLOCK_BYTES)
                        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, mfcc1024_kernel<false>, 256, G_BLOCK_BYTES);
    return e == hipSuccess && n > 0 ? n : 3;
}

}  // namespace dsp
