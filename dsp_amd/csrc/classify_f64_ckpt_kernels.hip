// classify_f64_ckpt_kernels.hip -- the float64 classifier of donut-classifier/classifier.c without materialised filter outputs.
//
// Round 3 wrote both filtered copies of every clip to HBM (2 x 128 KB per one-second clip beside 128 KB of input) and read them back
// for the transforms: 4.65 x the algorithmic traffic.  Here ONE pass over the input
//   iir2_screen_f64_kernel    runs both Butterworth recurrences (classifier.c:420-446, bit for bit: separate multiply and subtract in
//                             the reference's order) and stores only their RESTART STATES -- the delay line v[n-1 .. n-8] at offsets
//                             0, 64, 128, 192 of every spectrogram segment (64 B each; 18 KB per clip and filter instead of 128 KB) --
//                             and decides, for every segment of the 1000-3000 Hz output, find_midpoints' question "is a cell of this
//                             time bin above the threshold" (:679-745) WITHOUT a float64 transform: a screening DFT of the segment in
//                             bf16 on the matrix pipe with a rigorous error bound.  Loud for sure / quiet for sure are final; the
//                             segments in between (a cell within the bound of the threshold: ~0.06 dB for noise) go on a work list.
//   spec_f64_from_ckpt_kernel recomputes exactly the listed segments from their restart states -- the same operations on the same
//                             values in the same order as the whole-clip recurrence, so the same bits -- and transforms them in
//                             float64 (fft_frame, classify_f64_device.hpp): <flags> the undecided segments of the screening,
//                             <maps> every segment of the 3000-7500 Hz output of the clips that have midpoints.
// Midpoints and band sums follow as before (classify_f64_kernels.hip).  Every decision the classifier makes is made on float64 values:
// the screening only proves, segment by segment, what the float64 transform would have decided.
//
// The screening (why it is sound).  For a segment y[0..255] of the 1000-3000 Hz output, mean m, window w, the reference's cell of bin k
// is c_k |X_k|^2 / U, X_k = sum_n w_n (y_n - m) e^(-2 pi i k n / 256), c_k = 2 (1 for k = 0, 128).  The pass computes
//   D_k = sum_n (Whi + Wlo)_kn bf16(y_n),  W_kn = w_n e^(-2 pi i k n / 256) split into two bf16 tables (|Whi + Wlo - W| <= 2^-16 |W|)
// for k = 0 .. 63 and k = 128 with v_mfma_f32_32x32x16_bf16 (products exact in float32, float32 accumulation), then X~_k = D_k - m What_k
// (What = the window's own transform).  bf16(y) = y (1 + d), |d| <= 2^-8, so per component |X~_k - X_k| <= g := 1.03 * 2^-8 * sum_n |y_n|
// (the 3 % cover the table's 2^-16, the accumulation's 256 * 2^-24 and the float32 mean term), sqrt(2) g for the complex value.
//   a cell is loud for sure   when |X~_k| > T_k + sqrt2 g,  T_k = sqrt(U * threshold power / c_k)
//   quiet for sure            when |X~_k| < T_k - sqrt2 g
// and the bins 64 .. 127 that are not computed are bounded together by Parseval: sum_k c_k |X_k|^2 = 256 E, E = sum_n w_n^2 (y_n - m)^2,
// so every one of their cells is at most 256 E - sum_{computed k} c_k |X_k|^2, the subtracted sum taken low by the 2-norm of the
// rounding error (||e|| <= 1.1 * 16 * 2^-8 * sqrt(sum w_n^2 y_n^2): the transform of w * (bf16(y) - y) by Parseval again).  E, the sums
// and m come from the taps wave, which evaluates the filter's taps in float32 (its own error bound ev is added to g, ||e|| and E).  A segment is "quiet" only when every computed cell is quiet for
// sure AND that remainder is below the threshold; "loud" when one computed cell is loud for sure; otherwise it is listed.
//
// Block = 64 clips (lane = clip in the serial parts), four wavefronts -- one per SIMD, so that three blocks share a CU with one wave of
// each on every SIMD (five-wave blocks put two waves of a block on one SIMD and the third block of a CU did not fit beside the others):
//   R_bp  recurrence 3000-7500 Hz: x tile -> delay line, restart states to HBM           R_mp  the same at 1000-3000 Hz, v tile -> LDS
//         (the two also fetch the x tiles, three tiles ahead in registers)
//   T     taps of the 1000-3000 Hz filter one tile behind (y = b0 v + sum b_j v[n-j]), float32 sums per segment, y as bf16 -> LDS,
//         and the MFMAs + verdicts of table rows 0 .. 63 (bins 0 .. 31, 128) on that tile
//   X     two tiles behind: the MFMAs + verdicts of rows 64 .. 127 (bins 32 .. 63); writes the segments' flags
// A tile = 16 samples = one MFMA k-step = one full 128-byte line of a float64 row.  One __syncthreads per tile.
#include "diag_guard.hpp"
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "classify_f64_device.hpp"
#include "classify_kernels.hpp"

// classifier.c is compiled without contraction (plain gcc -O2 on x86-64): every product is rounded before it is added
#pragma clang fp contract(off)

namespace dsp {

using namespace f64dev;

namespace {

constexpr int SC_TS = 16;                         // samples per tile
constexpr int SC_XLD = 18;                        // doubles per LDS row of an x / v tile (144 B: 16-byte aligned rows, 36-dword stride)
constexpr int SC_YROW = 48;                       // bytes per clip row of a bf16 y tile (32 + 16: conflict-free ds_read_b128 / ds_write_b128)
constexpr int SC_RING = 4;                        // y tiles kept: the transform waves run two tiles behind and start a segment one tile late
constexpr int SC_THREADS = 256;
#ifndef SC_ROLES
#define SC_ROLES 7        // diagnostic builds: which roles are compiled in (1 recurrences, 2 taps, 4 transform waves)
#endif
constexpr int kTilesPerHop = kSpecHop / SC_TS;    // 14
constexpr int kTilesPerSeg = kSpecSeg / SC_TS;    // 16
constexpr int kCkStep = kCkStrideF64 / SC_TS;     // a restart state every 4 tiles
static_assert(kSpecHop % SC_TS == 0 && kSpecSeg % SC_TS == 0 && kCkStrideF64 % SC_TS == 0 && kSpecSeg == kCkPerSegF64 * kCkStrideF64, "tiles, segments and restart states line up");

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// ---- input kinds: what one sample is in HBM (IN of the kernels) ----
//   0 float64        1 int16 mono (s / 32768, classifier.c:55-59)       2 int16 interleaved stereo, channel 0 (:286-297)
//   3 int16 interleaved stereo, (L + R) / 65536 = the average of the two channels' s / 32768 (main_test.c:205-217 in double: exact)
template <int IN> struct In;
template <> struct In<0> { static constexpr int kBytes = 8, kPerPiece = 2; };     // bytes per sample (all channels), samples per 16-byte piece
template <> struct In<1> { static constexpr int kBytes = 2, kPerPiece = 8; };
template <> struct In<2> { static constexpr int kBytes = 4, kPerPiece = 4; };
template <> struct In<3> { static constexpr int kBytes = 4, kPerPiece = 4; };

// the samples of one 16-byte piece as doubles
template <int IN>
__device__ __forceinline__ void piece_to_f64(const u32x4 &q, double (&o)[In<IN>::kPerPiece])
{
    if constexpr (IN == 0) {
        o[0] = __hiloint2double((int)q[1], (int)q[0]);
        o[1] = __hiloint2double((int)q[3], (int)q[2]);
    } else if constexpr (IN == 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            o[2 * k] = (double)(int)(short)(q[k] & 0xffffu) * (1.0 / 32768.0);
            o[2 * k + 1] = (double)((int)q[k] >> 16) * (1.0 / 32768.0);
        }
    } else if constexpr (IN == 2) {
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = (double)(int)(short)(q[k] & 0xffffu) * (1.0 / 32768.0);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = (double)((int)(short)(q[k] & 0xffffu) + ((int)q[k] >> 16)) * (1.0 / 65536.0);
    }
}

// one sample read element-wise (rows that are not 16-byte aligned, pieces that cross the end of a row)
template <int IN>
__device__ __forceinline__ double sample_at(const void *__restrict__ x, long idx)
{
    if constexpr (IN == 0) return reinterpret_cast<const double *>(x)[idx];
    else if constexpr (IN == 1) return (double)reinterpret_cast<const short *>(x)[idx] * (1.0 / 32768.0);
    else if constexpr (IN == 2) return (double)reinterpret_cast<const short *>(x)[2 * idx] * (1.0 / 32768.0);
    else return (double)((int)reinterpret_cast<const short *>(x)[2 * idx] + (int)reinterpret_cast<const short *>(x)[2 * idx + 1]) * (1.0 / 65536.0);
}

}  // namespace

// =====================================================================================================================================
// pass 1: recurrences, restart states, screening
// =====================================================================================================================================
// RAGGED (clips of different lengths; spans[clip] = start, samples, segments): T_row is the row length of loud / the restart states
// (the longest clip's segment count), a block walks as many tiles as ITS longest clip has, and a lane stops keeping restart states and
// verdicts at its own clip's last segment; what it computes past that point (the next clip's samples; the last piece of the buffer
// again past its end) is never looked at.  Rows start anywhere: the 16-byte loads go out unaligned.
template <int IN, bool EVEN_B, bool VEC, bool RAGGED = false>
__global__ __launch_bounds__(SC_THREADS) __attribute__((amdgpu_waves_per_eu(3))) void iir2_screen_f64_kernel(const void *__restrict__ xin, long n_clips, int n, long stride, int T_row, const IirCoefD c_bp,
                                                                      const IirCoefD c_mp, double *__restrict__ ck_bp, double *__restrict__ ck_mp,
                                                                      const ScreenTablesD *__restrict__ tab, int *__restrict__ loud, int *__restrict__ want,
                                                                      float thr_u, float guard, int *__restrict__ cu_table,
                                                                      const ClipSpan *__restrict__ spans = nullptr, long total = 0)
{
    static_assert(!RAGGED || VEC, "ragged batches use the piece loads");
    __shared__ __attribute__((aligned(16))) double tin[2][64 * SC_XLD];                 // x tiles as doubles, [tile parity]
    __shared__ __attribute__((aligned(16))) double vbuf[2][64 * SC_XLD];                // v tiles of the 1000-3000 Hz filter
    __shared__ __attribute__((aligned(16))) unsigned char ybuf[SC_RING][64 * SC_YROW];  // its output as bf16, [tile mod 4][clip][16 samples]
    __shared__ float f_mean[64], f_g[64], f_e[64], f_en[64], f_sum[64];                  // per clip, of the segment that has just ended
    __shared__ unsigned f_state[64];
    __shared__ float what[65][2];                                                       // the window's transform at bins 0 .. 63, [64] = bin 128
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // Which wave runs which role (0 R_bp, 1 R_mp, 2 T, 3 X) follows the SIMD it landed on and the block's arrival number k on its CU.
    // The dispatcher gives the waves of a block to the four SIMDs in a fixed order, so with wave number = role the three blocks of a CU
    // stack their R_bp waves on one SIMD and their R_mp waves on another -- 768 float64 operations per step there, ~40 on the SIMD of
    // the X waves: measured 2.30 ms for the recurrences alone.  The table spreads them: per SIMD {R, R, X}, {R, R, X}, {R, T, T},
    // {R, T, X}.  Scheduling only: what a role computes does not depend on the wave that runs it; waves that share a SIMD (any other
    // placement) simply take the roles that are left.
    __shared__ int s_simd[4], s_role[4];
    {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xFu;     // HW_ID, XCC_ID
        if (lane == 0) s_simd[wib] = (int)((hw >> 4) & 3u);
        __syncthreads();
        if (threadIdx.x == 0) {
            int k = 0;
            if (cu_table) k = atomicAdd(cu_table + (((xcc << 8) | ((hw >> 8) & 0xFFu)) & (kSimdLoadCus - 1)), 1);      // (XCC, SE, SH, CU)
            constexpr int want_role[3][4] = {{0, 1, 2, 3}, {1, 3, 2, 0}, {3, 0, 1, 2}};      // [k mod 3][SIMD]
            bool taken[4] = {false, false, false, false};
            int role[4] = {-1, -1, -1, -1};
            for (int w = 0; w < 4; ++w) {
                const int r0 = want_role[k % 3][s_simd[w] & 3];
                if (cu_table && !taken[r0]) { role[w] = r0; taken[r0] = true; }
            }
            for (int w = 0; w < 4; ++w)
                if (role[w] < 0)
                    for (int r0 = 0; r0 < 4; ++r0)
                        if (!taken[r0]) { role[w] = r0; taken[r0] = true; break; }
            for (int w = 0; w < 4; ++w) s_role[w] = role[w];
#ifdef SC_DIAG
            if (cu_table) {
                cu_table[kSimdLoadCus + (long)blockIdx.x * 16] = (int)(((xcc << 8) | ((hw >> 8) & 0xFFu)) & (kSimdLoadCus - 1));
                cu_table[kSimdLoadCus + (long)blockIdx.x * 16 + 1] = k;
            }
#endif
        }
        __syncthreads();
    }
    const int wv = __builtin_amdgcn_readfirstlane(s_role[wib]);
#ifdef SC_DIAG      // diagnostic build: per block [16]: CU index, arrival number, the waves' SIMDs and roles, busy / total cycles per role
    unsigned long long diag_busy = 0, diag_t0 = 0;
    const unsigned long long diag_start = __builtin_amdgcn_s_memtime();
#define SC_DIAG_BEGIN() diag_t0 = __builtin_amdgcn_s_memtime()
#define SC_DIAG_END() diag_busy += __builtin_amdgcn_s_memtime() - diag_t0
#define SC_DIAG_WRITE()                                                                                                            \
    if (cu_table && lane == 0) {                                                                                                   \
        int *dg = cu_table + kSimdLoadCus + (long)blockIdx.x * 16;                                                                 \
        dg[4 + wv] = s_simd[wib];                                                                                                  \
        dg[8 + wv] = (int)(diag_busy >> 4);                                                                                        \
        dg[12 + wv] = (int)((__builtin_amdgcn_s_memtime() - diag_start) >> 4);                                                     \
    }
#else
#define SC_DIAG_BEGIN()
#define SC_DIAG_END()
#define SC_DIAG_WRITE()
#endif
    const long clip0 = (long)blockIdx.x * 64;
    const int rows = (int)((n_clips - clip0) < 64 ? (n_clips - clip0) : 64);
    __shared__ long s_base[RAGGED ? 64 : 1];
    __shared__ int s_T[RAGGED ? 64 : 1], s_Tmax;
    int T = T_row, my_T = T_row;                          // RAGGED: segments this block walks / of this lane's clip
    if (RAGGED) {
        if (threadIdx.x == 0) s_Tmax = 1;
        __syncthreads();
        if (threadIdx.x < 64) {
            ClipSpan sp{0, 0, 0};
            if ((int)threadIdx.x < rows) sp = spans[clip0 + threadIdx.x];
            s_base[threadIdx.x] = sp.off; s_T[threadIdx.x] = sp.frames;
            atomicMax(&s_Tmax, sp.frames);
        }
        __syncthreads();
        T = s_Tmax; my_T = s_T[lane];
    }
    const int n_tiles = (T - 1) * kTilesPerHop + kTilesPerSeg;           // samples past the last whole segment reach no output
    if (threadIdx.x < 64) {
        f_state[threadIdx.x] = 0u; f_sum[threadIdx.x] = 0.0f;
        what[threadIdx.x][0] = tab->what_re[threadIdx.x]; what[threadIdx.x][1] = tab->what_im[threadIdx.x];
        if (threadIdx.x == 0) { what[64][0] = tab->what128; what[64][1] = 0.0f; }
    }
#ifndef SC_PRIO
#define SC_PRIO 2
#endif
    if (SC_PRIO == 1) {            // recurrence waves first
        if (wv < 2) __builtin_amdgcn_s_setprio(3);
        else if (wv == 2) __builtin_amdgcn_s_setprio(1);
    } else if (SC_PRIO == 2) {     // short jobs first
        if (wv == 3) __builtin_amdgcn_s_setprio(3);
        else if (wv == 2) __builtin_amdgcn_s_setprio(1);
    }

    // ---- x tiles: the 128 threads of R_bp and R_mp fetch 128-byte row pieces and commit one compute tile per step ----
    // A queue of LD_DEPTH load tiles rides in registers: with one tile in flight per block (24 KB per CU) the float64 input could not
    // be streamed faster than ~3 TB/s at the loaded HBM latency, and the transform waves that used to load were the slowest of the block.
    constexpr int PP = In<IN>::kPerPiece;                 // samples per 16-byte piece
    constexpr int LT = 8 * PP;                            // samples per 128-byte load tile
    constexpr int KT = LT / SC_TS;                        // compute tiles per load tile (1, 4, 2)
    constexpr int LD_PER = 4;                             // pieces per loader thread and load tile (512 pieces over 128 threads)
    constexpr int LD_DEPTH = KT == 1 ? 3 : 2;             // load tiles in flight
    const int n_ltiles = (n_tiles + KT - 1) / KT;
    typedef u32x4 RawQueue[LD_DEPTH][LD_PER];        // (declared inside each loading role: its registers are that role's only)
    // VEC: every row start is 16-byte aligned -- 16-byte pieces.  Every load is UNCONDITIONAL on a clamped address (rows past the
    // block's last clip repeat that clip, pieces past the end of a row -- int16 input, last load tile: compute tiles past the last
    // segment, never committed -- read the row's last whole piece): a load under a condition lands in a temporary that is moved into the
    // queue's registers, and that move waits for the load just issued (the first form of this kernel exposed the whole HBM latency in
    // every step that way).  For the same reason the queue never rotates: slot = load tile mod LD_DEPTH, selected by a uniform switch.
    // !VEC (odd strides, unaligned bases): the commit reads its samples element by element, nothing is kept in flight.
    const long last_piece = ((long)n - PP) / PP * PP;     // first sample of the last whole piece of a row
    auto fetch = [&](u32x4 (&dst)[LD_PER], int L, int ltid) {          // thread takes piece c = ltid & 7 of rows (ltid >> 3) + 16 k
        if (!VEC) return;
        const int c = ltid & 7;
        long s0 = (long)L * LT + (long)c * PP;            // first sample of the piece
        s0 = s0 < last_piece ? s0 : last_piece;
#pragma unroll
        for (int k = 0; k < LD_PER; ++k) {
            int r = (ltid >> 3) + 16 * k;
            r = r < rows ? r : rows - 1;
            if (RAGGED) {
                long idx = s_base[r] + (long)L * LT + (long)c * PP;
                idx = idx < total - PP ? idx : total - PP;      // (past the buffer's end: its last whole piece again)
                dst[k] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(reinterpret_cast<const unsigned char *>(xin) + idx * In<IN>::kBytes));
            } else
            dst[k] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(reinterpret_cast<const unsigned char *>(xin) + ((clip0 + r) * stride + s0) * In<IN>::kBytes));
        }
    };
    auto commit = [&](const u32x4 (&src)[LD_PER], int cs, int ltid) {  // compute tile cs (of the load tile in src) -> tin[cs & 1]
        const int c = ltid & 7, sub = cs % KT;
        constexpr int PIECES_PER_TILE = SC_TS / PP;       // 8, 2, 4
        if (c / PIECES_PER_TILE != sub) return;
        const int off = (c % PIECES_PER_TILE) * PP;       // sample offset of the piece inside the compute tile
        double *dst = tin[cs & 1];
#pragma unroll
        for (int k = 0; k < LD_PER; ++k) {
            const int r = (ltid >> 3) + 16 * k;
            double o[PP];
            if (VEC) piece_to_f64<IN>(src[k], o);
            else {
                const long s0 = (long)(cs / KT) * LT + (long)c * PP;
                const void *row = reinterpret_cast<const unsigned char *>(xin) + (clip0 + (r < rows ? r : rows - 1)) * stride * In<IN>::kBytes;
#pragma unroll
                for (int j = 0; j < PP; ++j) o[j] = sample_at<IN>(row, s0 + j);
            }
#pragma unroll
            for (int j = 0; j < PP; ++j) dst[r * SC_XLD + off + j] = o[j];
        }
    };
    // during step s: compute tile s + 1 into LDS; when that uses its load tile up, the tile LD_DEPTH further on is requested into its slot
    // (slot_hint: the R / T loops are unrolled LD_DEPTH times, so that for float64 input -- one load tile per step -- the slot is a
    // compile-time constant of each copy and the compiler's vmcnt waits count exactly the younger loads; behind a run-time switch it
    // waits for every outstanding load.  int16 input changes slot every 4 / 2 steps and keeps the switch: a quarter of the bytes.)
    auto loader_step = [&](RawQueue &raw, int s, int ltid, int slot_hint) {
        const int cs = s + 1, L = cs / KT;
        // (the fetch is issued whether or not its tile exists -- past the end it re-reads the rows' last pieces: a fetch under a run-time
        // condition makes the number of younger loads unknown to the compiler, which then waits for all of them)
        const bool live = cs < n_tiles, used_up = cs % KT == KT - 1;
        (void)L;
        switch (slot_hint) {
        case 0: if (live) commit(raw[0], cs, ltid); if (used_up) fetch(raw[0], L + LD_DEPTH, ltid); break;
        case 1: if (live) commit(raw[1], cs, ltid); if (used_up) fetch(raw[1], L + LD_DEPTH, ltid); break;
        default: if (live) commit(raw[LD_DEPTH - 1], cs, ltid); if (used_up) fetch(raw[LD_DEPTH - 1], L + LD_DEPTH, ltid); break;
        }
    };
    auto loader_start = [&](RawQueue &raw, int ltid) {
#pragma unroll
        for (int dd = 0; dd < LD_DEPTH; ++dd) fetch(raw[dd], dd, ltid);
        loader_step(raw, -1, ltid, 0);
    };

    // Each role runs its own loop over the steps with ONE barrier per step (every wave of the block executes the same number of
    // barriers; the branch is wave-uniform): the compiler then allocates registers for the largest role, not for the sum of their states.
    // (round 4: int16 input too -- its slot changes every KT steps, so the loops are unrolled LD_DEPTH * KT times; behind the run-time switch
    // it kept, the int16 kernel was the slower one, 2.45 against 2.18 ms, with a quarter of the bytes)
    constexpr int UF = LD_DEPTH * KT;                     // steps after which the queue's slot pattern repeats
    const int n_steps = (n_tiles + 3 + UF - 1) / UF * UF;                       // (padded: the last steps only meet at the barrier)
    if ((SC_ROLES & 1) && wv < 2) {
        // ================= recurrence waves: tile s at step s =================
        const IirCoefD &c = wv == 0 ? c_bp : c_mp;
        double *ck = wv == 0 ? ck_bp : ck_mp;
        double d[8];                                      // v[n-1] .. v[n-8] of this lane's clip
#pragma unroll
        for (int j = 0; j < 8; ++j) d[j] = 0.0;
        RawQueue raw;
        loader_start(raw, wv * 64 + lane);
        __syncthreads();
        auto step = [&](int s, int slot_hint) {
            SC_DIAG_BEGIN();
            loader_step(raw, s, wv * 64 + lane, slot_hint);
            if (s < n_tiles) {
                const int seg = s / kTilesPerHop, p = s - seg * kTilesPerHop;
                // restart states: offsets 0, 64, 128, 192 of segment seg (tiles 14 seg + {0, 4, 8, 12}; the last two tiles of a
                // segment are the first two of the next)
                if (p % kCkStep == 0 && seg < my_T && lane < rows) {
                    d2 *dst = reinterpret_cast<d2 *>(ck + (((long)seg * kCkPerSegF64 + p / kCkStep) * n_clips + clip0 + lane) * 8);
#pragma unroll
                    for (int j = 0; j < 4; ++j) dst[j] = d2{d[2 * j], d[2 * j + 1]};      // (plain stores: the four 16-byte pieces of a lane's 64 bytes meet in L2; as nt stores they reached HBM as partial lines, 1.7 x the bytes)
                }
                const double *xrow = tin[s & 1] + lane * SC_XLD;
                double xr[SC_TS];
#pragma unroll
                for (int i = 0; i < SC_TS; i += 2) { const d2 v2 = *reinterpret_cast<const d2 *>(xrow + i); xr[i] = v2.x; xr[i + 1] = v2.y; }
#pragma unroll
                for (int i = 0; i < SC_TS; ++i) {                       // classifier.c:427-433
                    double v = xr[i];
#pragma unroll
                    for (int j = 1; j <= 8; ++j) v = v - c.a[j] * d[j - 1];
#pragma unroll
                    for (int j = 7; j > 0; --j) d[j] = d[j - 1];
                    d[0] = v;
                    xr[i] = v;
                }
                if (wv == 1) {
                    double *vrow = vbuf[s & 1] + lane * SC_XLD;
#pragma unroll
                    for (int i = 0; i < SC_TS; i += 2) *reinterpret_cast<d2 *>(vrow + i) = d2{xr[i], xr[i + 1]};
                }
            }
            SC_DIAG_END();
            __syncthreads();
        };
        // whole groups of LD_DEPTH steps, no condition between the copies (a copy that may be skipped makes its fetch conditional again)
        for (int s0 = 0; s0 < n_steps; s0 += UF) {
#pragma unroll
            for (int u = 0; u < UF; ++u) step(s0 + u, ((u + 1) / KT) % LD_DEPTH);      // the slot of load tile (s + 1) / KT
        }
        SC_DIAG_WRITE();
    } else if ((SC_ROLES & 6) && wv >= 2) {
        // ================= taps wave (tile s - 1 at step s) and transform wave (tile s - 2) =================
        const bool taps = wv == 2;
        const int xq = wv - 2;                            // table blocks 2 xq, 2 xq + 1 = bins 32 xq .. 32 xq + 31 (and bin 128 with xq = 0)
        // T: the taps in FLOAT32.  What they feed is the screening only (every float64 output of this filter that the classifier uses is
        // recomputed from the restart states): y32 = fl32(sum_j b_j v_j) on float32-rounded v differs from the float64 output by at most
        // 11 * 2^-24 * sum_j |b_j| |v[n-j]| (two roundings of the inputs, a chain of at most nine fused operations), which enters the
        // bounds below as ev = 2^-19 * (sum |b_j|) * (sum |v| over the segment and the eight samples before it) -- about 1e-5 of sum |y|
        // for these band-passes, beside bf16's 2^-8.  Nine float64 operations per sample became one conversion and six float32 ones on
        // the wave that shares its SIMD with two others.
        float bf[9], b_abs = 0.f;
#pragma unroll
        for (int j = 0; j < 9; ++j) { bf[j] = (float)c_mp.b[j]; b_abs += fabsf(bf[j]); }
        b_abs *= 1.0001f;
        float d[8];                                       // T: float32(v[n-1] .. v[n-8])
#pragma unroll
        for (int j = 0; j < 8; ++j) d[j] = 0.f;
        float s_cur = 0.f, a_cur = 0.f, q_cur = 0.f, s_prev = 0.f, a_prev = 0.f, q_prev = 0.f;       // T: sum y, sum |y|, sum w^2 y^2 of the open segments
        float v_cur = 0.f, v_prev = 0.f, v_tile = 0.f;    // T: sum |v| of the open segments (from one tile before their start) and of the last tile
        f32x16 acc[2][2];                                 // [table block][clip half], rows = (bin, re / im) pairs
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;
        int pending_seg = -1;                             // X: the segment whose verdicts are waiting in f_state / f_sum
        // (the lane number is made opaque once per step: everything derived from it -- LDS addresses of the verdict's table reads -- is
        // then recomputed where it is used instead of being kept in registers across the loop)
        int lane_o = lane;
        __syncthreads();                                  // (the loaders' first tile)
        for (int s = 0; s < n_steps; ++s) {
            SC_DIAG_BEGIN();
            asm volatile("" : "+v"(lane_o));
            const int r = lane_o & 31, h = lane_o >> 5;
            if (!taps && pending_seg >= 0) {
                // the verdicts of segment pending_seg: all four contributions (two lane halves x two waves) are in
                unsigned st = f_state[lane_o];                          // 0 every computed cell quiet for sure, 1 undecided, 2 a cell loud for sure
                if (st == 0u) {
                    const float root = sqrtf(f_sum[lane_o]) - f_en[lane_o];
                    const float low = root > 0.f ? root * root : 0.f;   // the computed bins hold at least this much of 256 E
                    if (!(f_e[lane_o] - low < thr_u * (1.0f - guard) * 0.9999f)) st = 1u;      // a bin that was not computed could reach the threshold
                }
                if (lane_o < rows && (!RAGGED || pending_seg < s_T[lane_o])) {
                    const long fr = (clip0 + lane_o) * T_row + pending_seg;
                    loud[fr] = st == 2u ? 1 : (st == 1u ? 2 : 0);
                    if (st == 1u) want[1 + atomicAdd(want, 1)] = (int)fr;
                }
                f_state[lane_o] = 0u; f_sum[lane_o] = 0.0f;
            }
            pending_seg = -1;
            const int tx = taps ? s - 1 : s - 2;
            // This tile's transform work as a short list of k-steps, ONE MFMA site and ONE verdict site in the code:
            //   p = 0: k-step 14 of segment tc - 1 (segment tc's first k-step waits: the accumulators still belong to tc - 1)
            //   p = 1: k-step 15 of segment tc - 1, its verdict, then k-steps 0 (the previous tile, still in the ring) and 1 of segment tc
            //   p > 1: k-step p of segment tc
            const bool tile_live = tx >= 0 && tx < n_tiles;
            const int tc = tile_live ? tx / kTilesPerHop : 0, p = tile_live ? tx - tc * kTilesPerHop : 0;
            const bool prev = tile_live && p < 2 && tc >= 1, cur = tile_live && tc < T && p >= 1;
            const int n_prev = prev ? 1 : 0, n_ops = (SC_ROLES & 4) ? n_prev + (cur ? (p == 1 ? 2 : 1) : 0) : 0;
            auto ks_of = [&](int o) { return o < n_prev ? kTilesPerHop + p : (p == 1 ? o - n_prev : p); };
            // the table fragments of the first k-step are requested HERE, before the taps: behind them their L2 latency (~2 000 cycles with
            // every CU streaming) stood in series with the taps and made this wave the block's slowest (85 % busy, the recurrences 53 - 67 %)
            bf16x8 a_pre[2][2];                           // [table block][hi / lo]
            auto request = [&](int ks) {
#pragma unroll
                for (int mbl = 0; mbl < 2; ++mbl) {
                    const bf16x8 *a = reinterpret_cast<const bf16x8 *>(tab->a_tab) + (size_t)((ks * 4 + 2 * xq + mbl) * 2) * 64 + lane_o;
                    a_pre[mbl][0] = a[0]; a_pre[mbl][1] = a[64];
                }
            };
            request(n_ops > 0 ? ks_of(0) : 0);
            if ((SC_ROLES & 2) && taps && tile_live) {
                const int ts = tx;                                      // segment tc at tile p; for p < 2 also segment tc - 1 at tile 14 + p
                if (p == 0) { s_prev = s_cur; a_prev = a_cur; q_prev = q_cur; s_cur = a_cur = q_cur = 0.f; v_prev = v_cur; v_cur = v_tile; }
                const double *vrow = vbuf[ts & 1] + lane_o * SC_XLD;
                float yf[SC_TS];
                v_tile = 0.f;
#pragma unroll
                for (int i = 0; i < SC_TS; i += 2) {                    // classifier.c:435-441 (in float32: see above)
                    const d2 v2 = *reinterpret_cast<const d2 *>(vrow + i);
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const float vf = (float)(e ? v2.y : v2.x);
                        float o = bf[0] * vf;
#pragma unroll
                        for (int j = 1; j <= 8; ++j)
                            if (!EVEN_B || j % 2 == 0) o = fmaf(bf[j], d[j - 1], o);
#pragma unroll
                        for (int j = 7; j > 0; --j) d[j] = d[j - 1];
                        d[0] = vf;
                        v_tile += fabsf(vf);
                        yf[i + e] = o;
                    }
                }
                v_cur += v_tile;
                if (p < 2) v_prev += v_tile;
                if (p >= 2) {                                           // the window is 1 here
#pragma unroll
                    for (int i = 0; i < SC_TS; ++i) { s_cur += yf[i]; a_cur += fabsf(yf[i]); q_cur = fmaf(yf[i], yf[i], q_cur); }
                } else {                                                // tapered in for segment tc, out for segment tc - 1
#pragma unroll
                    for (int i = 0; i < SC_TS; ++i) {
                        const float wi = tab->win2_in[SC_TS * p + i], wo = tab->win2_out[SC_TS * p + i];
                        s_cur += yf[i]; a_cur += fabsf(yf[i]); q_cur = fmaf(wi * yf[i], yf[i], q_cur);
                        s_prev += yf[i]; a_prev += fabsf(yf[i]); q_prev = fmaf(wo * yf[i], yf[i], q_prev);
                    }
                }
                // y as bf16, [clip][16 samples]: what an MFMA B fragment of 32 clips x 16 samples reads 16 bytes at a time
                u32x4 lo4, hi4;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    lo4[k] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{yf[2 * k], yf[2 * k + 1]}, bf16x2));
                    hi4[k] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{yf[8 + 2 * k], yf[9 + 2 * k]}, bf16x2));
                }
                u32x4 *yrow = reinterpret_cast<u32x4 *>(ybuf[ts & (SC_RING - 1)] + lane_o * SC_YROW);
                yrow[0] = lo4; yrow[1] = hi4;
                if (p == 1 && tc >= 1) {                                // segment tc - 1 is complete: what the verdicts need of it
                    const float m = s_prev * (1.0f / kSpecSeg);
                    const float ev = (1.0f / 524288.0f) * b_abs * v_prev;                            // float32 taps (see above)
                    const float e32 = q_prev + 2.0f * fabsf(m) * a_prev + m * m * tab->win2_sum;    // >= sum w^2 (y32 - m)^2
                    const float re = sqrtf(e32 * 1.0001f) + ev;                                      // >= sqrt(sum w^2 (y - m)^2)
                    f_mean[lane_o] = m;
                    f_g[lane_o] = 1.4142136f * (1.03f * (1.0f / 256.0f) * a_prev + ev);
                    f_e[lane_o] = (float)kSpecSeg * re * re * 1.0001f;
                    f_en[lane_o] = 16.0f * (1.1f * (1.0f / 256.0f) * sqrtf(q_prev) + ev);
                }
                wave_sync_lds();                                        // this wave reads the tile and the segment's figures back below
            }
            if (n_ops > 0) {
#pragma unroll 1
                for (int o = 0; o < n_ops; ++o) {
                    const bool is_prev = o < n_prev;
                    const int tile = (!is_prev && p == 1 && o == n_prev) ? tx - 1 : tx;
                    const unsigned char *yb = ybuf[tile & (SC_RING - 1)] + r * SC_YROW + 16 * h;
                    const bf16x8 b0 = *reinterpret_cast<const bf16x8 *>(yb), b1 = *reinterpret_cast<const bf16x8 *>(yb + 32 * SC_YROW);
#pragma unroll
                    for (int mbl = 0; mbl < 2; ++mbl) {
                        acc[mbl][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_pre[mbl][0], b0, acc[mbl][0], 0, 0, 0);
                        acc[mbl][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_pre[mbl][0], b1, acc[mbl][1], 0, 0, 0);
                        acc[mbl][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_pre[mbl][1], b0, acc[mbl][0], 0, 0, 0);
                        acc[mbl][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_pre[mbl][1], b1, acc[mbl][1], 0, 0, 0);
                    }
                    if (o + 1 < n_ops) request(ks_of(o + 1));           // (ahead of the verdict below, which does not need them)
                    if (is_prev && p == 1) {
                        // ---- the verdict on segment tc - 1, per clip (column) ----
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb) {
                            const int col = 32 * nb + r;
                            const float m = f_mean[col], g = f_g[col];
                            auto bounds = [&](float t2, float &hi2, float &lo2) {
                                const float t = sqrtf(t2), up = t + g, dn = t - g;
                                hi2 = fmaxf(up * up, t2 * (1.0f + guard)) * 1.00001f;
                                lo2 = dn > 0.f ? fminf(dn * dn, t2 * (1.0f - guard)) * 0.99999f : -1.0f;
                            };
                            float hi2, lo2;
                            bounds(0.5f * thr_u, hi2, lo2);             // |X|^2 at the threshold, c_k = 2
                            bool sure_loud = false, all_quiet = true;
                            float sum = 0.f;
#pragma unroll
                            for (int mbl = 0; mbl < 2; ++mbl)
#pragma unroll
                                for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                                    for (int e = 0; e < 2; ++e) {
                                        const int bin = 16 * (2 * xq + mbl) + 4 * gq + 2 * h + e;
                                        const f32x2 wh = *reinterpret_cast<const f32x2 *>(what[bin]);
                                        const float re = acc[mbl][nb][4 * gq + 2 * e] - m * wh.x;
                                        const float im = acc[mbl][nb][4 * gq + 2 * e + 1] - m * wh.y;
                                        if (mbl == 0 && gq == 0 && e == 0 && bin == 0) {
                                            // table row 1 holds bin 128 where bin 0 has its (zero) imaginary part: two real cells, c_k = 1
                                            float hi1, lo1;
                                            bounds(thr_u, hi1, lo1);
                                            const float r128 = acc[mbl][nb][1] - m * what[64][0];
                                            const float p0 = re * re, p128 = r128 * r128;
                                            sure_loud = sure_loud || p0 > hi1 || p128 > hi1;
                                            all_quiet = all_quiet && p0 < lo1 && p128 < lo1;
                                            sum += p0 + p128;
                                        } else {
                                            const float pw = re * re + im * im;
                                            sure_loud = sure_loud || pw > hi2;
                                            all_quiet = all_quiet && pw < lo2;
                                            sum = fmaf(2.0f, pw, sum);
                                        }
                                    }
                            atomicMax(&f_state[col], sure_loud ? 2u : (all_quiet ? 0u : 1u));
                            atomicAdd(&f_sum[col], sum);
                        }
                        pending_seg = tc - 1;
#pragma unroll
                        for (int a = 0; a < 2; ++a)
#pragma unroll
                            for (int b = 0; b < 2; ++b)
#pragma unroll
                                for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;
                    }
                }
            }
            SC_DIAG_END();
            __syncthreads();
        }
        SC_DIAG_WRITE();
    }
}

// =====================================================================================================================================
// recompute + float64 transform of listed segments
// =====================================================================================================================================
// A wavefront takes 8 work items (segments) at a time: lane (f, j) = (item, quarter), 32 lanes, loads the restart state of its quarter,
// runs the filter over its 64 samples -- the recurrence and the taps on each sample as classifier.c:427-441 writes them -- and leaves y
// in the item's LDS row; then the wave transforms the 8 rows two at a time, each IN PLACE in its own row (fft_frame_inplace).  x reaches
// the rows through coalesced 16-byte loads (a segment is 2 KB of contiguous float64), one pass ahead, and is overwritten by y.
// (First form of this round: 16 items per pass and fft_frame's two buffers -- 159 KB of LDS per block, ONE wave per SIMD, every phase
// bound by its own latencies: 1.6 ms for the 872 k segments of the bench's listed clips.  Half the items per pass leave half the lanes
// idle in the filter phase, which is bound by its dependent chain anyway; two waves per SIMD hide each other's latencies.)
//   MAPS = false  items = the frames on the work list want (want[0] entries, frame numbers clip * T + t): loud[frame] = 0 / 1
//   MAPS = true   items = all T segments of the clips on the work list hits: sxx[entry][t][129] = U * PSD
constexpr int RC_FRAMES = 8;                     // items per wave pass
constexpr int RC_CHUNK_LD = kCkStrideF64 + 2;     // doubles per quarter in an LDS row (+2: the four lanes of an item start on different banks)
constexpr int RC_ROW_LD = 296;                    // doubles per row: 2368 B = 64 mod 256, so that the 32 filter lanes (item f, quarter j: f x 2368 + j x 528 bytes) start on different banks
static_assert(RC_ROW_LD >= kCkPerSegF64 * RC_CHUNK_LD && RC_ROW_LD >= 2 * kInplaceCd && RC_ROW_LD % 2 == 0, "a row holds the four padded quarters / the transform's image and is 16-byte aligned");
template <bool MAPS, int IN, bool EVEN_B>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void spec_f64_from_ckpt_kernel(const void *__restrict__ xin, long n_clips, int n, long stride, int T, const IirCoefD c,
                                                                 const double *__restrict__ ck, const SpecTablesD *__restrict__ tab,
                                                                 const int *__restrict__ worklist, double *__restrict__ sxx, int *__restrict__ loud,
                                                                 double mid_power, double midpoint_db, double guard, int vec_ok,
                                                                 unsigned long long *__restrict__ minmax, const ClipSpan *__restrict__ spans = nullptr)
{
    // spans (ragged batches): clip c starts at spans[c].off; T stays the row length of the work lists and maps.  MAPS walks T slots per
    // listed clip: the slots past a clip's last segment take that last segment again (the same values to the same places).
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // [wave] rows[8][RC_ROW_LD] doubles, then the twiddles
    double *rows_all = reinterpret_cast<double *>(smem);
    FftTwiddles &tw = *reinterpret_cast<FftTwiddles *>(smem + (size_t)4 * RC_FRAMES * RC_ROW_LD * sizeof(double));
    // (wib through readfirstlane: the compiler must KNOW that a wave's item numbers are uniform, or the loop below is a divergent one)
    const int lane = threadIdx.x & 63, wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), half = lane >> 5, i = lane & 31;
    double *rows = rows_all + (size_t)wib * RC_FRAMES * RC_ROW_LD;
    fill_twiddles(tw, tab, threadIdx.x);
    __syncthreads();
    FftLane L;
    fft_lane_init(L, tw, tab, i);
    const double U = tab->U;
    const long n_entries = worklist[0];
    const long total = MAPS ? n_entries * T : n_entries;
    const long wave = (long)blockIdx.x * 4 + wib, n_waves = (long)gridDim.x * 4;
    const int f = (lane >> 2) & (RC_FRAMES - 1), j = lane & 3;          // recompute role (lanes 0 .. 31): item f of the pass, quarter j
    const bool filters = lane < 4 * RC_FRAMES;
    if (wave * RC_FRAMES >= total) return;
    // (clip, t) of item k of the pass that starts at item pass0.  MAPS: item = entry e * T + t, and (e0, t0) of a pass's first item is
    // CARRIED from pass to pass (constant step, one carry): round 4's first form divided a 64-bit item number by T for every 16-byte
    // piece it requested -- on the vector unit, because the wave number was not known to be uniform --, 21 emulated divisions per pass
    // of eight segments: a quarter of this kernel's instructions with float64 input (16 pieces per lane), 1.31 ms against 1.02 ms from int16.
    // Items past the end: the last one stands in (their results are not stored).
    auto locate = [&](long pass0, long e0, int t0, int k, long &clip, int &t) {
        if (MAPS) {
            long e = e0;
            t = t0 + k;
            while (t >= T) { t -= T; ++e; }
            if (e >= n_entries) { e = n_entries - 1; t = T - 1; }
            clip = worklist[1 + e];
            if (spans) { const int tc = spans[clip].frames; t = t < tc ? t : tc - 1; }
        } else {
            long it = pass0 + k;
            it = it < total ? it : total - 1;
            const int fr = worklist[1 + it];
            const int c = fr / T;
            clip = c; t = fr - c * T;
        }
    };
    auto base_of = [&](long clip) { return spans ? spans[clip].off : clip * stride; };
    const long pass_step = n_waves * RC_FRAMES;
    const long step_e = MAPS ? pass_step / T : 0;                        // once per wave
    const int step_t = MAPS ? (int)(pass_step - step_e * T) : 0;
    long e_req = MAPS ? (wave * RC_FRAMES) / T : 0;                      // the pass being REQUESTED starts at entry e_req, column t_req
    int t_req = MAPS ? (int)(wave * RC_FRAMES - e_req * T) : 0;
    long e_cur = 0;                                                      // ... and the pass being worked on
    int t_cur = 0;
    // A pass's input -- 16 segments of 256 samples as 16-byte pieces, coalesced (a segment is contiguous), and the restart state of
    // (item f, quarter j) -- is requested one pass AHEAD into registers: with one wave per SIMD (the rows fill the LDS) nothing else
    // hides the HBM latency, and four dependent load batches per pass were a quarter of this kernel's time.  Every load is
    // unconditional on a clamped item (the last one stands in for the items past the end; their results are not stored).
    constexpr int PP = In<IN>::kPerPiece;
    constexpr int PIECES = kSpecSeg / PP;                               // 16-byte pieces per segment: 128, 32, 64
    constexpr int PER_LANE = RC_FRAMES * PIECES / 64;                   // pieces per lane and pass: 32, 8, 16
    u32x4 q[PER_LANE];
    d2 ckq[4];
    auto request = [&](long item0) {
        if (vec_ok) {
#pragma unroll
            for (int u = 0; u < PER_LANE; ++u) {
                const int pc = lane + 64 * u, k = pc / PIECES, piece = pc % PIECES;
                long clip; int t;
                locate(item0, e_req, t_req, k, clip, t);
                const long s0 = (long)t * kSpecHop + (long)piece * PP;
                q[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(reinterpret_cast<const unsigned char *>(xin) + (base_of(clip) + s0) * In<IN>::kBytes));
            }
        }
        long clip; int t;
        locate(item0, e_req, t_req, f, clip, t);
        const d2 *src = reinterpret_cast<const d2 *>(ck + (((long)t * kCkPerSegF64 + j) * n_clips + clip) * 8);
#pragma unroll
        for (int u = 0; u < 4; ++u) ckq[u] = src[u];
        // the request's position becomes the working one; the next request lies one pass further
        e_cur = e_req; t_cur = t_req;
        e_req += step_e; t_req += step_t;
        if (t_req >= T) { t_req -= T; ++e_req; }
    };
    request(wave * RC_FRAMES);
    for (long item0 = wave * RC_FRAMES; item0 < total; item0 += n_waves * RC_FRAMES) {
        // ---- x into the rows ----
#pragma unroll
        for (int u = 0; u < PER_LANE; ++u) {
            const int pc = lane + 64 * u, k = pc / PIECES, piece = pc % PIECES;
            double o[PP];
            if (vec_ok) piece_to_f64<IN>(q[u], o);
            else {
                long clip; int t;
                locate(item0, e_cur, t_cur, k, clip, t);
                const void *row = reinterpret_cast<const unsigned char *>(xin) + base_of(clip) * In<IN>::kBytes;
                const long s0 = (long)t * kSpecHop + (long)piece * PP;
#pragma unroll
                for (int jj = 0; jj < PP; ++jj) o[jj] = sample_at<IN>(row, s0 + jj);
            }
            const int n0 = piece * PP;                                   // first sample of the piece inside the segment
            double *dst = rows + k * RC_ROW_LD + (n0 / kCkStrideF64) * RC_CHUNK_LD + n0 % kCkStrideF64;
#pragma unroll
            for (int jj = 0; jj < PP; ++jj) dst[jj] = o[jj];
        }
        // ---- restart state of (item f, quarter j) ----
        double d[8];
#pragma unroll
        for (int u = 0; u < 4; ++u) { d[2 * u] = ckq[u].x; d[2 * u + 1] = ckq[u].y; }
        // ---- the next pass's input on its way while this one is filtered and transformed ----
        const long e_now = e_cur;                                        // (request() moves the working position on)
        const int t_now = t_cur;
        request(item0 + n_waves * RC_FRAMES);
        wave_sync_lds();
        // ---- the filter over the quarter: classifier.c:427-441 per sample, y over x in place ----
        if (filters) {
            double *row = rows + f * RC_ROW_LD + j * RC_CHUNK_LD;
#pragma unroll 1
            for (int h0 = 0; h0 < kCkStrideF64; h0 += 16) {
                double xr[16];
#pragma unroll
                for (int u = 0; u < 16; u += 2) { const d2 v2 = *reinterpret_cast<const d2 *>(row + h0 + u); xr[u] = v2.x; xr[u + 1] = v2.y; }
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    double v = xr[u];
#pragma unroll
                    for (int a = 1; a <= 8; ++a) v = v - c.a[a] * d[a - 1];
                    double o = c.b[0] * v;
#pragma unroll
                    for (int a = 1; a <= 8; ++a)
                        if (!EVEN_B || a % 2 == 0) o = o + c.b[a] * d[a - 1];
#pragma unroll
                    for (int a = 7; a > 0; --a) d[a] = d[a - 1];
                    d[0] = v;
                    xr[u] = o;
                }
#pragma unroll
                for (int u = 0; u < 16; u += 2) *reinterpret_cast<d2 *>(row + h0 + u) = d2{xr[u], xr[u + 1]};
            }
        }
        wave_sync_lds();
        // ---- transforms: rows 2 turn + half ----
        for (int turn = 0; turn < RC_FRAMES / 2; ++turn) {
            const int k = 2 * turn + half;
            const long it = item0 + k;
            if (item0 + 2 * turn >= total) break;                        // wave-uniform: neither half has a segment
            double *row = rows + k * RC_ROW_LD;
            d2 x[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = *reinterpret_cast<const d2 *>(row + r * RC_CHUNK_LD + 2 * i);
            wave_sync_lds();                                             // the row becomes the transform's second buffer
            double m[4], m128;
            fft_frame_inplace(x, L, tw, reinterpret_cast<cd *>(row), i, m, m128);
            if (MAPS) {
                if (it < total) {
                    double *out = sxx + it * (long)kSpecBins;
#pragma unroll
                    for (int r = 0; r < 4; ++r) out[i + 32 * r] = m[r];
                    if (i == 0) out[128] = m128;
                }
                if (minmax) {
                    // the clip's smallest / largest positive cell (classifier.c:105-125 needs them of the whole map): this frame's, over
                    // the 32 lanes of the half, then one atomic pair -- the band kernel no longer scans the map it reads windows of
                    double lo = __longlong_as_double(0x7FF0000000000000ll), hi = 0.0;
                    const double c4 = i == 0 ? m128 : 0.0;
#pragma unroll
                    for (int r = 0; r < 5; ++r) {
                        const double v = r < 4 ? m[r] : c4;
                        if (v > 0) { lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
                    }
                    auto red = [&](auto ctrl_tag) {
                        constexpr int CTRL = decltype(ctrl_tag)::value;
                        const double l2 = dpp_f64<CTRL>(lo), h2 = dpp_f64<CTRL>(hi);
                        lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi;
                    };
                    red(std::integral_constant<int, 0xB1>{}); red(std::integral_constant<int, 0x4E>{});
                    red(std::integral_constant<int, 0x141>{}); red(std::integral_constant<int, 0x140>{});
                    {   // the neighbouring row of 16 lanes (v_permlane16_swap)
                        const int llo = __double2loint(lo), lhi = __double2hiint(lo), hlo = __double2loint(hi), hhi = __double2hiint(hi);
                        const auto a = __builtin_amdgcn_permlane16_swap(llo, llo, false, false), b = __builtin_amdgcn_permlane16_swap(lhi, lhi, false, false);
                        const auto c2 = __builtin_amdgcn_permlane16_swap(hlo, hlo, false, false), d2_ = __builtin_amdgcn_permlane16_swap(hhi, hhi, false, false);
                        const double l0 = __hiloint2double(b[0], a[0]), l1 = __hiloint2double(b[1], a[1]);
                        const double h0 = __hiloint2double(d2_[0], c2[0]), h1 = __hiloint2double(d2_[1], c2[1]);
                        lo = l0 < l1 ? l0 : l1; hi = h0 > h1 ? h0 : h1;
                    }
                    if (i == 0 && it < total && hi > 0) {
                        long clip; int t;
                        locate(item0, e_now, t_now, k, clip, t);
                        atomicMin(&minmax[2 * clip], (unsigned long long)__double_as_longlong(lo));
                        atomicMax(&minmax[2 * clip + 1], (unsigned long long)__double_as_longlong(hi));
                    }
                }
            } else {
                const bool hit = frame_is_loud(m, m128, i, half, U, mid_power, midpoint_db, guard);
                if (i == 0 && it < total) loud[worklist[1 + it]] = hit;
            }
            wave_sync_lds();
        }
    }
}

// =====================================================================================================================================
// host side
// =====================================================================================================================================
namespace {

unsigned short bf16_rne(float v)
{
    unsigned u;
    std::memcpy(&u, &v, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
float bf16_to_float(unsigned short b)
{
    const unsigned u = (unsigned)b << 16;
    float v;
    std::memcpy(&v, &u, 4);
    return v;
}

bool even_taps_only(const IirCoefD &c) { return c.b[1] == 0.0 && c.b[3] == 0.0 && c.b[5] == 0.0 && c.b[7] == 0.0; }

int columns_of(int n) { return n < kSpecSeg ? 0 : (n - kSpecSeg) / kSpecHop + 1; }

template <typename K>
int resident_blocks(K kernel, int threads, size_t smem)      // per device: the persistent grids walk their work from there
{
    int dev = 0, cus = 0, per = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, kernel, threads, smem) != hipSuccess || cus <= 0 || per <= 0) {
        (void)hipGetLastError();
        cus = 256; per = 1;
    }
    return cus * per;
}

// alignment the 16-byte loads need: every row start and every piece on 16 bytes
template <int IN>
bool rows_vec_ok(const void *x, long stride)
{
    return reinterpret_cast<uintptr_t>(x) % 16 == 0 && (stride * In<IN>::kBytes) % 16 == 0;
}

}  // namespace

bool build_screen_tables_f64(const SpecTablesD &spec, int fs, ScreenTablesD &t)
{
    // the taper is 32 samples at either end (classifier.c:504-521 with alpha 0.25: n <= 32 and n >= 224) and the window is exactly 1 in
    // between: the taps wave takes the squares of the first / last two tiles from the table and 1 elsewhere
    for (int n = kSpecSeg - kSpecHop; n < kSpecHop; ++n)
        if (spec.win[n] != 1.0) return false;
    double w2 = 0.0;
    for (int n = 0; n < kSpecSeg; ++n) w2 += spec.win[n] * spec.win[n];
    if (std::fabs(w2 * fs - spec.U) > 1e-9 * spec.U) return false;
    t.win2_sum = (float)(w2 * (1.0 + 1e-6));
    for (int n = 0; n < kSpecSeg - kSpecHop; ++n) {
        t.win2_in[n] = (float)(spec.win[n] * spec.win[n] * (1.0 + 1e-6));                 // (rounded up: they enter upper bounds only)
        t.win2_out[n] = (float)(spec.win[kSpecHop + n] * spec.win[kSpecHop + n] * (1.0 + 1e-6));
    }
    const long double PI2 = 6.283185307179586476925286766559005768L;
    auto row_value = [&](int row, int n) -> double {                     // table row -> (bin, part): rows 2 q, 2 q + 1 = re, im of bin q; row 1 = bin 128
        const int bin = row == 1 ? 128 : row >> 1;
        const long double a = PI2 * (long double)((bin * n) & 255) / 256.0L;
        return spec.win[n] * (double)((row & 1) && row != 1 ? -sinl(a) : cosl(a));
    };
    for (int ks = 0; ks < 16; ++ks)
        for (int mb = 0; mb < 4; ++mb)
            for (int l = 0; l < 64; ++l)
                for (int jj = 0; jj < 8; ++jj) {
                    const double v = row_value(32 * mb + (l & 31), 16 * ks + 8 * (l >> 5) + jj);
                    const unsigned short hi = bf16_rne((float)v);
                    const unsigned short lo = bf16_rne((float)(v - (double)bf16_to_float(hi)));
                    t.a_tab[ks][mb][0][l][jj] = hi;
                    t.a_tab[ks][mb][1][l][jj] = lo;
                }
    for (int bin = 0; bin < 64; ++bin) {
        long double re = 0, im = 0;
        for (int n = 0; n < kSpecSeg; ++n) {
            const long double a = PI2 * (long double)((bin * n) & 255) / 256.0L;
            re += spec.win[n] * cosl(a);
            im -= spec.win[n] * sinl(a);
        }
        t.what_re[bin] = (float)re;
        t.what_im[bin] = (float)im;
    }
    long double s128 = 0;
    for (int n = 0; n < kSpecSeg; ++n) s128 += (n & 1) ? -spec.win[n] : spec.win[n];
    t.what128 = (float)s128;
    return true;
}

int f64_screen_blocks_per_pass()
{
    // the blocks of a pass must all be resident (every block runs the whole clip length); 64 clips per block
    static int per_device[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return 256; }
    if (per_device[dev] == 0) per_device[dev] = resident_blocks(iir2_screen_f64_kernel<0, true, true>, SC_THREADS, 0);
    return per_device[dev];
}

template <int IN>
static hipError_t launch_screen_in(const void *x, long n_clips, int n, long stride, const IirCoefD &c_bp, const IirCoefD &c_mp, double *ck_bp,
                                   double *ck_mp, const ScreenTablesD *tables, double U, double midpoint_db, double guard, int *loud, int *want,
                                   int *cu_table, hipStream_t stream, const ClipSpan *spans, long total)
{
    const int T = columns_of(n);
    const int blocks = (int)((n_clips + 63) / 64);
    const double mid_power = 1e-12 * std::pow(10.0, midpoint_db / 10.0);
    const float thr_u = (float)(mid_power * U);
    const float g = (float)std::min(std::max(guard, 0.0), 0.999);
    const bool vec = rows_vec_ok<IN>(x, stride), even = even_taps_only(c_mp);
#define DSP_SC_LAUNCH(E, V)                                                                                                                      \
    hipLaunchKernelGGL((iir2_screen_f64_kernel<IN, E, V>), dim3(blocks), dim3(SC_THREADS), 0, stream, x, n_clips, n, stride, T, c_bp, c_mp, ck_bp, \
                       ck_mp, tables, loud, want, thr_u, g, cu_table)
    if (spans) {                  // ragged batches: the designed (even-tap) numerators, unaligned piece loads
        if (!even || total < In<IN>::kPerPiece) return hipErrorInvalidValue;
        hipLaunchKernelGGL((iir2_screen_f64_kernel<IN, true, true, true>), dim3(blocks), dim3(SC_THREADS), 0, stream, x, n_clips, n, stride, T, c_bp, c_mp, ck_bp,
                           ck_mp, tables, loud, want, thr_u, g, cu_table, spans, total);
        return hipGetLastError();
    }
    if (even && vec) DSP_SC_LAUNCH(true, true);
    else if (even) DSP_SC_LAUNCH(true, false);
    else if (vec) DSP_SC_LAUNCH(false, true);
    else DSP_SC_LAUNCH(false, false);
#undef DSP_SC_LAUNCH
    return hipGetLastError();
}

hipError_t launch_iir2_screen_f64(const void *x, int in_kind, long n_clips, int n, long stride, const IirCoefD &c_bp, const IirCoefD &c_mp,
                                  double *ck_bp, double *ck_mp, const ScreenTablesD *tables, double U, double midpoint_db, double guard,
                                  int *loud, int *want, int *cu_table, hipStream_t stream, const ClipSpan *spans, long total)
{
    const int T = columns_of(n);
    if (n_clips <= 0 || T <= 0) return hipSuccess;
    if (n_clips * (long)T >= (1L << 31)) return hipErrorInvalidValue;                 // frame numbers are ints
    if ((n_clips + 63) / 64 > f64_screen_blocks_per_pass()) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(want, 0, sizeof(int), stream);
    if (e != hipSuccess) return e;
    // DSP_AMD_F64_ROLES=0: wave number = role (A/B runs)
    static const bool spread = [] { const char *v = std::getenv("DSP_AMD_F64_ROLES"); return !(v && std::atoi(v) == 0); }();
    if (!spread) cu_table = nullptr;
    if (cu_table && (e = hipMemsetAsync(cu_table, 0, sizeof(int) * kSimdLoadCus, stream)) != hipSuccess) return e;
    switch (in_kind) {
    case 0: return launch_screen_in<0>(x, n_clips, n, stride, c_bp, c_mp, ck_bp, ck_mp, tables, U, midpoint_db, guard, loud, want, cu_table, stream, spans, total);
    case 1: return launch_screen_in<1>(x, n_clips, n, stride, c_bp, c_mp, ck_bp, ck_mp, tables, U, midpoint_db, guard, loud, want, cu_table, stream, spans, total);
    case 2: return launch_screen_in<2>(x, n_clips, n, stride, c_bp, c_mp, ck_bp, ck_mp, tables, U, midpoint_db, guard, loud, want, cu_table, stream, spans, total);
    case 3: return launch_screen_in<3>(x, n_clips, n, stride, c_bp, c_mp, ck_bp, ck_mp, tables, U, midpoint_db, guard, loud, want, cu_table, stream, spans, total);
    default: return hipErrorInvalidValue;
    }
}

namespace {

constexpr size_t kRcSmem = (size_t)4 * RC_FRAMES * RC_ROW_LD * sizeof(double) + sizeof(FftTwiddles);

template <bool MAPS, int IN, bool EVEN_B>
hipError_t launch_rc(const void *x, long n_clips, int n, long stride, const IirCoefD &c, const double *ck, const SpecTablesD *tab, const int *worklist,
                     double *sxx, int *loud, double mid_power, double midpoint_db, double guard, hipStream_t stream, unsigned long long *minmax, const ClipSpan *spans)
{
    auto kernel = spec_f64_from_ckpt_kernel<MAPS, IN, EVEN_B>;
    static bool attr_set[64] = {false};
    static int resident[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRcSmem);
        if (e != hipSuccess) return e;
        resident[dev] = resident_blocks(kernel, 256, kRcSmem);
        attr_set[dev] = true;
    }
    const int T = columns_of(n);
    const long max_items = n_clips * (long)T;                            // the bound: the list's count is read on the device
    const long blocks = std::min<long>((max_items + 4 * RC_FRAMES - 1) / (4 * RC_FRAMES), resident[dev]);
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(256), kRcSmem, stream, x, n_clips, n, stride, T, c, ck, tab, worklist, sxx, loud, mid_power,
                       midpoint_db, guard, spans ? 1 : (int)rows_vec_ok<IN>(x, stride), minmax, spans);
    return hipGetLastError();
}

template <bool MAPS, int IN>
hipError_t launch_rc_b(const void *x, long n_clips, int n, long stride, const IirCoefD &c, const double *ck, const SpecTablesD *tab, const int *worklist,
                       double *sxx, int *loud, double mid_power, double midpoint_db, double guard, hipStream_t stream, unsigned long long *minmax, const ClipSpan *spans)
{
    return even_taps_only(c) ? launch_rc<MAPS, IN, true>(x, n_clips, n, stride, c, ck, tab, worklist, sxx, loud, mid_power, midpoint_db, guard, stream, minmax, spans)
                             : launch_rc<MAPS, IN, false>(x, n_clips, n, stride, c, ck, tab, worklist, sxx, loud, mid_power, midpoint_db, guard, stream, minmax, spans);
}

template <bool MAPS>
hipError_t launch_rc_k(const void *x, int in_kind, long n_clips, int n, long stride, const IirCoefD &c, const double *ck, const SpecTablesD *tab,
                       const int *worklist, double *sxx, int *loud, double mid_power, double midpoint_db, double guard, hipStream_t stream,
                       unsigned long long *minmax = nullptr, const ClipSpan *spans = nullptr)
{
    switch (in_kind) {
    case 0: return launch_rc_b<MAPS, 0>(x, n_clips, n, stride, c, ck, tab, worklist, sxx, loud, mid_power, midpoint_db, guard, stream, minmax, spans);
    case 1: return launch_rc_b<MAPS, 1>(x, n_clips, n, stride, c, ck, tab, worklist, sxx, loud, mid_power, midpoint_db, guard, stream, minmax, spans);
    case 2: return launch_rc_b<MAPS, 2>(x, n_clips, n, stride, c, ck, tab, worklist, sxx, loud, mid_power, midpoint_db, guard, stream, minmax, spans);
    case 3: return launch_rc_b<MAPS, 3>(x, n_clips, n, stride, c, ck, tab, worklist, sxx, loud, mid_power, midpoint_db, guard, stream, minmax, spans);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace

hipError_t launch_spec_f64_recheck(const void *x, int in_kind, long n_clips, int n, long stride, const IirCoefD &c_mp, const double *ck_mp,
                                   const SpecTablesD *tables, const int *want, double midpoint_db, double guard, int *loud, hipStream_t stream, const ClipSpan *spans)
{
    if (n_clips <= 0 || columns_of(n) <= 0) return hipSuccess;
    const double mid_power = 1e-12 * std::pow(10.0, midpoint_db / 10.0);
    return launch_rc_k<false>(x, in_kind, n_clips, n, stride, c_mp, ck_mp, tables, want, nullptr, loud, mid_power, midpoint_db, guard, stream, nullptr, spans);
}

hipError_t launch_spec_f64_listed_from_ckpt(const void *x, int in_kind, long n_clips, int n, long stride, const IirCoefD &c_bp, const double *ck_bp,
                                            const SpecTablesD *tables, const int *hits, double *sxx, hipStream_t stream, unsigned long long *minmax, const ClipSpan *spans)
{
    if (n_clips <= 0 || columns_of(n) <= 0) return hipSuccess;
    return launch_rc_k<true>(x, in_kind, n_clips, n, stride, c_bp, ck_bp, tables, hits, sxx, nullptr, 0.0, 0.0, 0.0, stream, minmax, spans);
}

}  // namespace dsp
