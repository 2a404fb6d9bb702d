// mfcc512_pair_kernel.hip -- the reference shape (512-sample frames, 40 mel, 13 coefficients, independent frames) with TWO
// frames per wavefront step.  Experiment of round 2 (DSP_KERNEL_PAIR): fewer VALU instructions per frame than
// mfcc512_wave_kernel (no v_permlane exchange, radix-8 butterflies), the same LDS traffic; see DESIGN.md 3 for what it measured.
//
// A frame's 512-point real FFT is a 256-point complex one; 256 = 4 x 64.  Two frames u, w sit in the eight register slots of
// a lane (slots 0-3: u[l + 64 a], slots 4-7: w[l + 64 a]); each gets its own radix-4 butterfly over a and the twiddle
// W256^(l q').  What is left is eight independent 64-point transforms over the lane index -- exactly what the radix-8
// pipeline of mfcc1024_wave_kernel.hip does for its eight first-stage outputs: its two LDS exchanges and two radix-8 stages run
// here unchanged, with slot q = q' + 4 f.  The frames never mix: no value of one frame enters an operation of the other.
// Afterwards lane l holds bins j + 32 r (r = 0..7) of frame f = (l >> 2) & 1, j = (l & 3) + 4 (l >> 3).
// Tail (untangling, power, sparse mel, 16-frame tile with log + DCT on v_mfma_f32_16x16x4_f32): as in mfcc_kernels.hip.
#include "diag_guard.hpp"
#include <hip/hip_runtime.h>

#include "mfcc_device.hpp"
#include "mfcc_kernels.hpp"
#include "tables.hpp"

#ifndef DSP_PAIR_WAVES
#define DSP_PAIR_WAVES 4
#endif
#ifndef DSP_PAIR_DIAG
#define DSP_PAIR_DIAG 0
#endif
#ifndef DSP_PAIR_PRIO
#define DSP_PAIR_PRIO 2          // 0 none; 1: 0 / 1 / 2 / 3 rising through the step; 2: 0 until stage C, then 2 / 3
#endif
#define PAIR_PRIO(a, b) do { if (DSP_PAIR_PRIO == 1) __builtin_amdgcn_s_setprio(a); else if (DSP_PAIR_PRIO == 2) __builtin_amdgcn_s_setprio(b); } while (0)

namespace dsp {

namespace {

constexpr int P_ZBUF = 0;                    // 512 x float2 exchange image; later P of frame 0 at float 0, of frame 1 at float P_PB1
constexpr int P_PART = 4096;                 // per frame 64 partial sums + the slot that reads 0 (65 floats each)
constexpr int P_ETILE = P_PART + 528;        // mel energies of 16 frames, E[mel][frame ^ (mel >> 2)]
constexpr int P_TILE_ROWS = 40;               // the reference's 40 mel filters (4 P_KS)
constexpr int P_WAVE_BYTES = P_ETILE + P_TILE_ROWS * 16 * 4;
static_assert(P_WAVE_BYTES % 16 == 0, "keep the carve 16-byte aligned");
constexpr int P_PB1 = 544;                   // P of frame 1 starts here (floats): 32 banks away from frame 0's, so the two halves of a wave's store do not collide
constexpr int P_KS = 10;                     // MFMA k-steps (4 mel filters each): the reference's 40 filters

// forward radix-8 butterfly: u[q] = sum_a v[a] W8^(a q)   (as in mfcc1024_wave_kernel.hip)
__device__ __forceinline__ void radix8p(c32 (&v)[8])
{
    c32 e[4] = {v[0], v[2], v[4], v[6]}, o[4] = {v[1], v[3], v[5], v[7]};
    radix4(e);
    radix4(o);
    constexpr float R2 = 0.70710678118654752f;
    const c32 t1 = {(o[1].x + o[1].y) * R2, (o[1].y - o[1].x) * R2};
    const c32 t2 = {o[2].y, -o[2].x};
    const c32 t3 = {(o[3].y - o[3].x) * R2, -(o[3].x + o[3].y) * R2};
    v[0] = cadd(e[0], o[0]); v[4] = csub(e[0], o[0]);
    v[1] = cadd(e[1], t1);   v[5] = csub(e[1], t1);
    v[2] = cadd(e[2], t2);   v[6] = csub(e[2], t2);
    v[3] = cadd(e[3], t3);   v[7] = csub(e[3], t3);
}

}  // namespace

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DSP_PAIR_WAVES))) void mfcc512_pair_kernel(const Mfcc512Args args,
                                                                                                             const PairExtra512 *__restrict__ X)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char *wl = smem + wib * P_WAVE_BYTES;
    float2 *zbuf = reinterpret_cast<float2 *>(wl + P_ZBUF);
    float *pbuf = reinterpret_cast<float *>(wl + P_ZBUF);
    float *part = reinterpret_cast<float *>(wl + P_PART);
    float *etile = reinterpret_cast<float *>(wl + P_ETILE);
    float *a_lds = reinterpret_cast<float *>(smem + 4 * P_WAVE_BYTES);           // MFMA A operand, block-shared
    float2 *tw2_lds = reinterpret_cast<float2 *>(a_lds + P_KS * 64);            // [p][lane], p < 7: second-stage twiddles
    float2 *twp_lds = tw2_lds + 7 * 64;                                          // [t][lane], t < 4: untangling twiddles
    const LaneTables512 *__restrict__ T = args.tables;

    // ---- per-lane constants ------------------------------------------------------------------------------------------
    float win[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) win[i] = T->win[i][lane];
    c32 twa[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) twa[q] = {T->tw1[2 * q][lane], T->tw1[2 * q + 1][lane]};          // W256^(l q')
    // the second-stage and the untangling twiddles (22 VGPRs) come from a block-shared LDS copy, 11 ds_read_b64 per pair: that
    // is the difference between three and four waves per SIMD
    for (int i = threadIdx.x; i < 7 * 64; i += 256) tw2_lds[i] = make_float2(X->tw2[2 * (i >> 6)][i & 63], X->tw2[2 * (i >> 6) + 1][i & 63]);
    for (int i = threadIdx.x; i < 4 * 64; i += 256) twp_lds[i] = make_float2(X->twp[2 * (i >> 6)][i & 63], X->twp[2 * (i >> 6) + 1][i & 63]);
    float melw[kMelChunk];
#pragma unroll
    for (int i = 0; i < kMelChunk; ++i) melw[i] = T->mel_w[i][lane];
    const int mel_k0 = T->mel_k0[lane];
    int gat[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) gat[g] = T->mel_src[g][lane];
    for (int i = threadIdx.x; i < P_KS * 64; i += 256) a_lds[i] = (&T->dct_a[0][0][0])[i];
    __syncthreads();
    const int n_mels = args.n_mels, n_mfcc = args.n_mfcc;
    const int l_hi = lane >> 3, l_lo = lane & 7;
    const int fr = (lane >> 2) & 1;                               // the frame of the pair this lane's bins belong to
    const int jj = (lane & 3) + 4 * (lane >> 3);                  // its bins: jj + 32 r
    const int partner = X->partner[lane] << 2;                    // byte index for ds_bpermute
    const bool self_paired = jj == 0;
    if (lane == 0) { part[kZeroSlot] = 0.0f; part[65 + kZeroSlot] = 0.0f; }
    wave_lds_sync();

    const long wave = (long)blockIdx.x * 4 + wib;
    const long n_waves = (long)gridDim.x * 4;
    const unsigned amin_u = __float_as_uint(args.amin);
    const float neg_top_db = -args.top_db;
    const long n_frames = args.n_frames;
    const float *__restrict__ in = static_cast<const float *>(args.in);

    // one cursor, one pair of look-ahead.  chunk is a multiple of 16: a pair never straddles chunks, only the batch's last
    // frame can be alone
    WaveCursor<false> pre;
    pre.init(wave, n_waves, args.chunk, n_frames, 0, 512, 0);
    if (!pre.valid()) return;
    c32 nx[2][4];
    long f_next = 0;
    bool two_next = false;
    auto fetch = [&]() {
        f_next = pre.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const f2v x = __builtin_nontemporal_load(reinterpret_cast<const f2v *>(in + pre.off + 2 * (lane + 64 * a)));
            nx[0][a] = {x.x, x.y};
        }
        pre.next();
        two_next = pre.valid();
        if (two_next) {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const f2v x = __builtin_nontemporal_load(reinterpret_cast<const f2v *>(in + pre.off + 2 * (lane + 64 * a)));
                nx[1][a] = {x.x, x.y};
            }
            pre.next();
        } else {
#pragma unroll
            for (int a = 0; a < 4; ++a) nx[1][a] = {0.0f, 0.0f};
        }
    };
    fetch();

    // ---- 16-frame tile epilogue: per-frame max, log, DCT on the matrix core (mfcc.c:169-216) --------------------------------
    int slot = 0;
    long fb0 = 0, fb1 = 0;
    auto flush = [&](int count) {
        wave_lds_sync();
        const int n = lane & 15, q = lane >> 4;
        unsigned mx = amin_u;
#pragma unroll 2
        for (int s = 0; s < P_KS; ++s) mx = max(mx, __float_as_uint(etile[64 * s + 16 * q + (n ^ s)]));
        {
            auto r = __builtin_amdgcn_permlane16_swap(mx, mx, false, false);
            mx = max(r[0], r[1]);
            r = __builtin_amdgcn_permlane32_swap(mx, mx, false, false);
            mx = max(r[0], r[1]);
        }
        const float rinv = __builtin_amdgcn_rcpf(__uint_as_float(mx));
        f4v acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
        auto kstep = [&](int s, f4v &acc) {
            const float ec = __uint_as_float(max(__float_as_uint(etile[64 * s + 16 * q + (n ^ s)]), amin_u));
            float db = 3.01029995663981195f * __builtin_amdgcn_logf(ec * rinv);
            db = __builtin_amdgcn_fmed3f(db, neg_top_db, 0.0f);
            if (4 * s + q >= n_mels) db = 0.0f;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a_lds[s * 64 + lane], db, acc, 0, 0, 0);
        };
#pragma unroll 1
        for (int s = 0; s < P_KS; s += 2) { kstep(s, acc0); kstep(s + 1, acc1); }
        const f4v d = acc0 + acc1;
        const long fl = (n < 8 ? fb0 : fb1 - 8) + n;
        const bool ok = n < count && fl < n_frames;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = 4 * q + j;
            if (ok && c < n_mfcc) args.out[fl * n_mfcc + c] = d[j];
        }
        wave_lds_sync();
    };

    while (true) {
        const long f0 = f_next;
        const bool two = two_next;
        PAIR_PRIO(0, 0);
        c32 v[8];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            v[a] = {nx[0][a].x * win[2 * a], nx[0][a].y * win[2 * a + 1]};
            v[4 + a] = {nx[1][a].x * win[2 * a], nx[1][a].y * win[2 * a + 1]};
        }
        const bool more = pre.valid();
#if DSP_PAIR_DIAG == 2
        if (more) { f_next = pre.f; pre.next(); two_next = pre.valid(); if (two_next) pre.next(); }      // timing build: arithmetic only
#else
        if (more) fetch();
#endif

        // ---- stage A': one radix-4 butterfly per frame over a, twiddle W256^(l q') ------------------------------------------
        {
            c32 u[4] = {v[0], v[1], v[2], v[3]}, w[4] = {v[4], v[5], v[6], v[7]};
            radix4(u);
            radix4(w);
#pragma unroll
            for (int q = 1; q < 4; ++q) { u[q] = cmul(u[q], twa[q - 1]); w[q] = cmul(w[q], twa[q - 1]); }
#pragma unroll
            for (int q = 0; q < 4; ++q) { v[q] = u[q]; v[4 + q] = w[q]; }
        }
        // exchange 1: element (q, l_hi, l_lo) at (64 q + 8 l_hi + l_lo) ^ 8 (q & 3): writer lane (l_hi, l_lo) slot q,
        // reader lane (q, l_lo) slot l_hi
#pragma unroll
        for (int q = 0; q < 8; ++q) zbuf[(64 * q + lane) ^ (8 * (q & 3))] = make_float2(v[q].x, v[q].y);
        wave_lds_sync();
#pragma unroll
        for (int h = 0; h < 8; ++h) {
            const float2 x = zbuf[(64 * l_hi + 8 * h + l_lo) ^ (8 * (l_hi & 3))];
            v[h] = {x.x, x.y};
        }
        wave_lds_sync();
        PAIR_PRIO(1, 0);
        // ---- stage B: radix-8 over l_hi, twiddle W64^(l_lo p) -----------------------------------------------------------
        radix8p(v);
#pragma unroll
        for (int p = 1; p < 8; ++p) {
            const float2 w = tw2_lds[64 * (p - 1) + lane];
            v[p] = cmul(v[p], c32{w.x, w.y});
        }
        // exchange 2: element (q, p, l_lo) at (64 p + 8 q + l_lo) ^ ((p & 3) << 1 | q >> 2): writer lane (q, l_lo) slot p,
        // reader lane (p, q) slot l_lo -> after stage C lane l holds pipeline bins l + 64 r
#pragma unroll
        for (int p = 0; p < 8; ++p) zbuf[(64 * p + lane) ^ (((p & 3) << 1) | (l_hi >> 2))] = make_float2(v[p].x, v[p].y);
        wave_lds_sync();
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const float2 x = zbuf[(64 * l_hi + 8 * l_lo + m) ^ (((l_hi & 3) << 1) | (l_lo >> 2))];
            v[m] = {x.x, x.y};
        }
        wave_lds_sync();
        PAIR_PRIO(2, 2);
        // ---- stage C: radix-8 over l_lo: v[r] = Z_f[jj + 32 r] / 2 ----------------------------------------------------------
        radix8p(v);

        // ---- untangle + power: pairs (k, 256 - k), k = jj + 32 t, t = 0..3, inside the frame's 32 lanes ---------------------
        float P[8];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            c32 b;
            b.x = __int_as_float(__builtin_amdgcn_ds_bpermute(partner, __float_as_int(v[7 - t].x)));
            b.y = __int_as_float(__builtin_amdgcn_ds_bpermute(partner, __float_as_int(v[7 - t].y)));
            const c32 own = v[(8 - t) & 7];                       // jj = 0: Z[256 - 32 t] is this lane's own slot (8 - t) % 8
            b.x = self_paired ? own.x : b.x;
            b.y = self_paired ? own.y : b.y;
            const c32 a = v[t];
            const c32 E = {a.x + b.x, a.y - b.y};
            const c32 O = {a.x - b.x, a.y + b.y};
            const float2 wt = twp_lds[64 * t + lane];
            const c32 Tw = cmul(O, c32{wt.x, wt.y});
            const float xr = E.x + Tw.y, xi = E.y - Tw.x;
            const float mr = E.x - Tw.y, mi = E.y + Tw.x;
            P[2 * t] = xr * xr + xi * xi;
            P[2 * t + 1] = mr * mr + mi * mi;
        }
        const float p128 = 4.0f * (v[4].x * v[4].x + v[4].y * v[4].y);          // jj = 0: bin 128 pairs with itself
        {
            float *pb = pbuf + P_PB1 * fr;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                pb[jj + 32 * t] = P[2 * t];
                pb[256 - jj - 32 * t] = P[2 * t + 1];
            }
            if (self_paired) pb[128] = p128;
        }
        PAIR_PRIO(3, 3);
        wave_lds_sync();

        // ---- sparse mel filterbank, both frames (mfcc.c:158-164) ---------------------------------------------------------
#pragma unroll
        for (int fi = 0; fi < 2; ++fi) {
            const float *rd = pbuf + P_PB1 * fi + mel_k0;
            float acc = 0.0f;
#pragma unroll
            for (int i = 0; i < kMelChunk; ++i) acc = fmaf(melw[i], rd[i], acc);
            part[65 * fi + lane] = acc;
        }
        wave_lds_sync();
        float e[2];
#pragma unroll
        for (int fi = 0; fi < 2; ++fi) {
            const float *pt = part + 65 * fi;
            float s = pt[gat[0]];
            s += pt[gat[1]];
            s += pt[gat[2]];
            e[fi] = lane < n_mels ? s : 0.0f;
        }
        if ((slot & 7) == 0) { if (slot == 0) fb0 = f0; else fb1 = f0; }
        if (lane < P_TILE_ROWS) {
            etile[16 * lane + (slot ^ (lane >> 2))] = e[0];
            if (two) etile[16 * lane + ((slot + 1) ^ (lane >> 2))] = e[1];
        }
        slot += two ? 2 : 1;
        if (slot >= 16 || !more) { flush(slot); slot = 0; }
        else wave_lds_sync();
        if (!more) return;
    }
}

hipError_t launch_mfcc512_pair(const Mfcc512Args &args, const PairExtra512 *extra, int blocks, hipStream_t stream)
{
    if (args.frames_per_clip != 0 || args.frame_len != 512 || args.in_kind != 0 || args.chunk % 16 != 0 || args.log_mode != 0 ||
        args.n_mels > 4 * P_KS || args.n_mfcc > 16)
        return hipErrorInvalidConfiguration;
    const size_t lds = (size_t)4 * P_WAVE_BYTES + (size_t)P_KS * 64 * 4 + (size_t)11 * 64 * 8;
    hipLaunchKernelGGL(mfcc512_pair_kernel, dim3(blocks), dim3(256), lds, stream, args, extra);
    return hipGetLastError();
}

int mfcc512_pair_blocks_per_cu()
{
    int n = 0;
    const size_t lds = (size_t)4 * P_WAVE_BYTES + (size_t)P_KS * 64 * 4 + (size_t)11 * 64 * 8;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, mfcc512_pair_kernel, 256, lds);
    return e == hipSuccess && n > 0 ? n : 3;
}

}  // namespace dsp
