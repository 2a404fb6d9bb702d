// tables.hpp -- host-side constant tables for the MFCC kernels.
//
// The reference ships its constants as generated headers (mfcc_params.h); the
// generator is 2fa/audio/word/python/export_mfcc_params.py.  Here the same
// formulas are evaluated at plan-creation time for any configuration and then
// re-laid-out per lane for the wave-per-frame kernel.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/dsp_amd.h"

// Exchange implementation of the last two FFT stage boundaries (A/B switches,
// see DESIGN.md "FFT exchanges"): 1 = through the per-wave LDS tile with XOR
// swizzles, 0 = DPP register moves.  The final lane -> bin map depends on exchange 3.
#ifndef DSP_X2_LDS
#define DSP_X2_LDS 1
#endif
#ifndef DSP_X3_LDS
#define DSP_X3_LDS 1
#endif

namespace dsp {

// ---- plain tables (reference layouts) --------------------------------------
std::vector<float> make_window(int kind, int n);                                  // export_mfcc_params.py:46
std::vector<float> make_frame_window(const dsp_mfcc_config &cfg);                 // frame_length taps; win_length centred (librosa pad_center)
std::vector<float> make_mel_filterbank(int sample_rate, int n_fft, int n_mels,    // :49-57, [n_mels][n_fft/2+1]
                                       float fmin, float fmax, int mel_norm);
std::vector<float> make_dct_ortho(int n_mfcc, int n_mels);                        // :27-41, [n_mfcc][n_mels]

// ---- per-lane layout for the 512-point wave-per-frame kernel ---------------
// Every array is [field][64 lanes] so a wave loads one field with one coalesced
// dword load per lane.
constexpr int kLanes = 64;
constexpr int kMelChunk = 12;   // bins per lane in the sparse mel product
constexpr int kMelGather = 6;   // max chunks per filter (kernels are instantiated for 3 and 6)
constexpr int kDctMaxLen = 20;  // log-mel values per lane in the DCT (n_mels / split, padded even)
constexpr int kDctSteps = 16;  // MFMA k-steps of 4 mel filters (n_mels <= 64)
constexpr int kZeroSlot = 64;   // LDS partial slot that always reads 0

struct LaneTables512 {
    // window (x0.5, see kernel) for samples 2(l+64a) and 2(l+64a)+1, a = 0..3
    float win[8][kLanes];
    // FFT twiddles, (cos, sin) pairs for q = 1..3
    float tw1[6][kLanes];   // W256^(l q)
    float tw2[6][kLanes];   // W64^((l%16) q)
    float tw3[6][kLanes];   // W16^((l%4) q)
    float twp[4][kLanes];   // W512^kappa, W512^(kappa+64); kappa(l) = the bin (mod 64) lane l ends up with
    int32_t kappa[kLanes];  // l (exchange 3 through LDS) or 16*(l&3) + 4*((l>>2)&3) + (l>>4) (DPP)
    int32_t partner[kLanes];// lane that holds bin (64 - kappa) mod 64
    // sparse mel: lane owns bins [k0, k0+12) of one filter
    int32_t mel_k0[kLanes];
    float mel_w[kMelChunk][kLanes];
    // filter m (lane m) = sum of up to mel_gather partial slots
    int32_t mel_src[kMelGather][kLanes];
    int32_t mel_gather;        // 3 or 6: chunks the widest filter needs, rounded up
    int32_t mel_conflict_free; // 1 when the window starts of each 32-lane half are distinct mod 32
    // DCT: `split` lanes cooperate on one coefficient; lane = split*c + q
    // multiplies log-mel [q*len, (q+1)*len)
    float dct_w[kDctMaxLen][kLanes];
    int32_t dct_split;         // 4 (n_mfcc <= 16) or 2 (n_mfcc <= 32)
    int32_t dct_len;           // floats per part (even, <= kDctMaxLen)
    int32_t n_mels, n_mfcc;
    // DCT as the A operand of v_mfma_f32_16x16x4_f32 (16-frame tile epilogue): lane (c = l%16, q = l/16),
    // k-step s holds D[16 ct + c][4 s + q] (0 outside n_mfcc x n_mels)
    float dct_a[2][kDctSteps][kLanes];
};

// ---- extra per-lane constants of the two-frames-per-wave kernel (mfcc512_pair_kernel.hip) ------------------------------------
// Two 512-sample frames ride the radix-8 pipeline of the 1024-point kernel as eight independent 64-point transforms: slots
// q = q' + 4 f (f = frame, q' = output of the frame's first radix-4 butterfly).  Lane l ends with bins j + 32 r (r = 0..7) of
// frame f = (l >> 2) & 1, j = (l & 3) + 4 (l >> 3).
struct PairExtra512 {
    float tw2[14][kLanes];      // W64^((l % 8) p), p = 1..7 (cos, sin): after the second stage
    float twp[8][kLanes];       // W512^(j + 32 t), t = 0..3: untangling the packed real transform
    int32_t partner[kLanes];    // lane of the same frame that holds bins (32 - j) + 32 r
};
void build_pair_extra_512(PairExtra512 &t);

// ---- per-lane layout of the FFT front end of the row-per-frame kernel -------------
// (4 frames per wave: lane = 16 g + j, row g = frame, 16 complex points per lane; the
// tail -- mel, log, DCT -- reuses LaneTables512).
struct RowTables512 {
    float win[32][kLanes];   // x0.5 window for samples 2(j+16k), 2(j+16k)+1 at [2k], [2k+1]
    float tw[30][kLanes];    // W256^(j q), q = 1..15: (cos, sin) at [2(q-1)], [2(q-1)+1]
    float twp[16][kLanes];   // W512^(j+16m), m = 0..7: (cos, sin) at [2m], [2m+1]
};
void build_row_tables_512(const dsp_mfcc_config &cfg, RowTables512 &t);

// ---- tables of the general 1024-point kernel (mfcc1024_kernel.hip) ------------------
constexpr int kGenChunks = 4;      // mel chunks (12 bins) per lane  -> up to 256 chunks
constexpr int kGenGather = 6;      // partial sums per filter (filters up to 72 bins wide)
constexpr int kGenMelsPerLane = 2; // filters per lane -> n_mels <= 128
constexpr int kGenDctLen = 32;     // log-mel values per DCT lane (split 4)
constexpr int kGenDctSteps = 32;   // MFMA k-steps of 4 mel filters (n_mels <= 128)
struct GenTables1024 {
    float win[16][kLanes];                 // x0.5 window for samples 2(l+64a), 2(l+64a)+1
    float w512[2][512];                    // W512^i (cos, sin): stage twiddles
    float w1024[2][256];                   // W1024^k (cos, sin), k < 256: untangling
    int32_t mel_k0[kGenChunks][kLanes];    // chunk c of lane l reads P[k0 .. k0+12)
    float mel_w[kGenChunks][kMelChunk][kLanes];
    int32_t mel_src[kGenMelsPerLane][kGenGather][kLanes];   // filter m = lane + 64 i sums these partial slots
    float dct_w[kGenDctLen][kLanes];       // lane 4c+q: row c, mels [32q, 32q+32)
    int32_t n_chunk_slots;                 // chunk slots per lane in use (1..4)
    int32_t n_mels, n_mfcc;
    // per-lane tables of the register-resident wave kernel (mfcc1024_wave_kernel.hip), all (cos, sin) pairs of exp(-2 pi i x):
    float tw1[14][kLanes];                 // W512^(l q),        q = 1..7 at [2(q-1)], [2(q-1)+1]: after the first radix-8 stage
    float tw2[14][kLanes];                 // W64^((l % 8) p),   p = 1..7: after the second stage
    float twp[8][kLanes];                  // W1024^(l + 64 t),  t = 0..3: untangling the packed real transform
    float dct_a[kGenDctSteps][kLanes];     // MFMA A operand of the tile epilogue: lane (c = l % 16, q = l / 16), step s: D[c][4 s + q]
    float win_chunk[kLanes][16];           // the x0.5 window in the prefilter's chunk order: lane l, samples 16 l .. 16 l + 15 (full 1024-sample frames)
};
constexpr int kGenZeroSlot = kGenChunks * kLanes;   // partial slot that always reads 0
bool build_gen_tables_1024(const dsp_mfcc_config &cfg, GenTables1024 &t, std::string &why);

// ---- n_fft = 2048 (the framing of cepstrum/scrubjay_infer.c:10-14: WIN_SIZE 2048, HOP_SIZE 1024, 40 filters, 20 coefficients)
// Tables of mfcc2048_kernel.hip: window per lane, twiddle tables, the mel filterbank as one run of non-zero weights per filter
// (CSR), the DCT rows.
constexpr int k2048MaxMels = 128, k2048MaxMfcc = 32, k2048MaxWeights = 4096;
constexpr int k2048SegTaps = 16, k2048SegSlots = 3, k2048MaxGather = 12;   // mel product in segments of <= 16 bins, <= 12 per filter
constexpr int k2048SegZero = k2048SegSlots * kLanes;                       // partial slot that always reads 0
struct GenTables2048 {
    float win[32][kLanes];                 // x0.5 window for samples 2(l+64a), 2(l+64a)+1 at [2a], [2a+1], a < 16
    float w1024[2][1024];                  // W1024^i (cos, sin): stage twiddles of the 1024-point complex FFT
    float w2048[2][512];                   // W2048^k (cos, sin), k < 512: untangling the packed real transform
    int32_t mel_lo[k2048MaxMels], mel_len[k2048MaxMels], mel_off[k2048MaxMels];   // filter m: bins [lo, lo+len), weights at mel_w[off ..]
    float mel_w[k2048MaxWeights];
    float dct[k2048MaxMfcc][k2048MaxMels]; // DCT-II rows
    int32_t n_mels, n_mfcc;
    // the same filterbank cut into segments of <= 16 consecutive bins, one per (slot, lane): lane l of slot s dots
    // P[seg_k0 .. seg_k0 + 16) with seg_w; filter m then adds its partial sums [mel_s0, mel_s0 + mel_cnt) in ascending order.
    // seg_ok = 0 when the bank does not fit 192 segments / 12 per filter: the kernel walks the CSR rows above instead.
    int32_t seg_ok;
    int32_t seg_k0[k2048SegSlots][kLanes];
    float seg_w[k2048SegSlots][k2048SegTaps][kLanes];
    int32_t mel_s0[k2048MaxMels], mel_cnt[k2048MaxMels];
    // DCT rows as the kernel's lanes read them: lane 2 c + h dots log-mels [h half, h half + half), half = ceil(n_mels / 2),
    // with dct_t[i][lane] = D[c][h half + i] (0 past the row / past n_mfcc)
    float dct_t[(k2048MaxMels + 1) / 2][kLanes];
};
bool build_gen_tables_2048(const dsp_mfcc_config &cfg, GenTables2048 &t, std::string &why);

// ---- per-frame Butterworth prefilter as a wave-parallel scan (BASELINE config 3 inside the 1024-point kernel) ----------
// The 8th-order filter H(z) = B(z^-1) / A(z^-1) (donut-classifier/classifier.c:342-401) in PARALLEL FORM: a direct term
// plus four second-order sections, one per conjugate pole pair,
//     H = k0 + sum_s (b0_s + b1_s z^-1) / (1 + a1_s z^-1 + a2_s z^-2),
// so that the filter state splits into four independent 2-vectors and a frame can be filtered by 64 lanes at once: lane l
// runs samples [16 l, 16 l + 16) from zero state, a Kogge-Stone scan over the lanes carries the section states across the
// chunk boundaries (state_out = M_s^16 state_in + local; step d multiplies by pw[d][s] = M_s^(16 * 2^d), M_s = [[-a1, -a2],
// [1, 0]]), and the lane reruns its chunk from its true initial state.  float64 throughout; equals the direct-form-II
// recurrence to ~1e-13 (checked at plan creation on 1024 random samples), i.e. to the last bit or so of the float result.
struct PrefilterScan {
    double k0;
    double b0[4], b1[4], a1[4], a2[4];
    double pw[6][4][4];            // [step][section][m00, m01, m10, m11]
    // Scan steps section s needs: after `steps[s]` steps the scan covers 2^steps[s] chunks back, and what lies further back
    // reaches the state damped by |pole_s|^(16 * 2^steps[s]) < 1e-14 (the pole radii of the two literal filters are
    // 0.27 .. 0.93: 1 .. 5 steps instead of 6 for every section).
    int32_t steps[4];
    // ---- the same filter as a CASCADE of four second-order sections (round 3; the form the 1024-point kernel runs) -------------
    // H = g prod_s (1 - z^-2) / (1 + a1_s z^-1 + a2_s z^-2): the literal numerators are g (1 - z^-2)^4 exactly (checked), the pole
    // pairs are sorted by radius (c_a1 / c_a2, widest damping first).  Each section is a lane-parallel scan of its own (chunk from
    // zero state, Kogge-Stone over the lanes with c_pw[d][s] = M_s^(16 * 2^d), chunk again from its true state).  A cascade has
    // no cancellation BETWEEN sections (the parallel form's four terms cancel to -76 dB in the stop band, which is why it needs
    // float64 throughout), so only the first two sections -- the ones that see the unattenuated stop-band energy -- run in
    // float64 and the last two in float32 (c_*f: the same constants rounded to float).  Measured on a float32 emulation of
    // exactly this lane structure against the float64 direct form under the MFCC gate (tools/emulate_prefilter_cascade.py):
    // worst 1e-5 of the gate's 1e-4 on stop-band-only frames, 1e-6 typical; all-float32 fails those frames (1.7e-4).
    double c_gain;
    double c_a1[4], c_a2[4];
    double c_pw[6][4][4];
    float c_a1f[4], c_a2f[4];
    float c_pwf[6][4][4];
    int32_t c_steps[4];            // scan steps per section: float64 sections to 1e-13, float32 sections to 1e-9
    int32_t c_ok;                  // 0: the numerator is not g (1 - z^-2)^4 -> the kernel runs the parallel form
    // Row form of the same scan (the kernel's DSP_PRE_ROWSCAN): Kogge-Stone steps of 1, 2, 4, 8 lanes INSIDE each 16-lane row
    // (DPP row_shr moves: no LDS crossbar, lanes without a source read 0), then one "cross-row" step per row that still matters:
    // lane (r, j) adds c_rowm[s][j] = M_s^(16 (j + 1)) times the value of lane 15 of row r - 1 (DPP row_bcast:15).  After the first
    // such step lane 15 of a row holds its row's total plus M^256 times the previous row's, so a second step reaches two rows back.
    double c_rowm[4][16][4];
    float c_rowmf[4][16][4];
    int32_t c_row_ok;              // the row form reproduces the direct form (self-check of the algebra in double)
};
// 1: mfcc1024_wave_kernel<PRE> runs the row form (plans need c_row_ok); 0: the wave-wide Kogge-Stone scan (A/B builds)
#ifndef DSP_PRE_ROWSCAN
#define DSP_PRE_ROWSCAN 1
#endif
// steps inside a row and cross-row steps of the row form for a section whose lane scan needs `steps` Kogge-Stone steps
constexpr int scan_row_steps(int steps) { return steps < 4 ? steps : 4; }
constexpr int scan_row_rounds(int steps) { return steps <= 4 ? 1 : steps - 3; }      // 5 -> 2 (32 chunks back), 6 -> 3 (the whole wave)
constexpr int kScanChunk = 16;     // samples per lane: 64 lanes x 16 = one 1024-sample frame
bool build_prefilter_scan(const double b[9], const double a[9], PrefilterScan &out, std::string &why);

// Fills `t`; returns false (with a message) when the configuration does not
// fit this kernel's layout.
bool build_lane_tables_512(const dsp_mfcc_config &cfg, LaneTables512 &t, std::string &why);

}  // namespace dsp
