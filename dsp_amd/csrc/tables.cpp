// tables.cpp -- see tables.hpp.
#include "tables.hpp"

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstring>

namespace dsp {

static const double kPi = 3.14159265358979323846;

// scipy.signal.get_window(name, n, fftbins=True) (export_mfcc_params.py:46):
// periodic form, float64 evaluation, one rounding to float32.
std::vector<float> make_window(int kind, int n)
{
    std::vector<float> w(n);
    for (int i = 0; i < n; ++i) {
        const double ph = 2.0 * kPi * i / n;
        double v = 1.0;
        if (kind == DSP_WINDOW_HANN) v = 0.5 - 0.5 * std::cos(ph);
        else if (kind == DSP_WINDOW_HAMMING) v = 0.54 - 0.46 * std::cos(ph);
        w[i] = (float)v;
    }
    return w;
}

// The per-frame window: `frame_length` taps; a shorter `win_length` window sits centred with
// zeros either side (librosa.util.pad_center, 2fa/audio/word/python/keyword_classifier.py:42-55).
std::vector<float> make_frame_window(const dsp_mfcc_config &cfg)
{
    if (cfg.win_length <= 0 || cfg.win_length >= cfg.frame_length) return make_window(cfg.window, cfg.frame_length);
    std::vector<float> w(cfg.frame_length, 0.0f);
    const std::vector<float> core = make_window(cfg.window, cfg.win_length);
    const int lpad = (cfg.frame_length - cfg.win_length) / 2;
    for (int i = 0; i < cfg.win_length; ++i) w[lpad + i] = core[i];
    return w;
}

// DSP_MELNORM_AUBIO_SLANEY: the bank new_aubio_mfcc(win_s, 40, ...) installs (cepstrum/scrubjay_infer.c:30 ->
// aubio_filterbank_set_mel_coeffs_slaney, aubio 0.4 src/spectral/filterbank_mel.c): Malcolm Slaney's Auditory Toolbox
// edges -- 13 filters spaced 66.67 Hz from 133.33 Hz, then 27 spaced by the factor 1.0711703, 42 edges in all, computed in
// float like aubio's smpl_t -- and triangles of UNIT AREA (height 2 / (upper - lower)) sampled at the bin frequencies
// k * sample_rate / n_fft.  aubio fills the rows with three loops (skip to the first bin above `lower`, rise while the NEXT
// bin is below `center`, fall while the next bin is below `upper`, never touching the Nyquist bin); bin for bin that is
// min(rise, fall) clipped at 0 of the triangle's two lines, which is what is evaluated here.  Parity unpinned (aubio is
// not vendored by the reference): tests/ check it against a loop-for-loop restatement and a float64 formula.
static std::vector<float> make_aubio_slaney_filterbank(int sample_rate, int n_fft)
{
    const int n_bins = n_fft / 2 + 1, n_filters = 40;
    float edge[42];
    for (int i = 0; i < 13; ++i) edge[i] = 133.3333f + (float)i * 66.66666666f;
    for (int i = 0; i < 29; ++i) edge[13 + i] = edge[12] * std::pow(1.0711703f, (float)(i + 1));
    const float bin_hz = (float)sample_rate / (float)n_fft;
    std::vector<float> fb((size_t)n_filters * n_bins, 0.0f);
    for (int m = 0; m < n_filters; ++m) {
        const float lo = edge[m], ce = edge[m + 1], hi = edge[m + 2];
        const float height = 2.0f / (hi - lo);
        const float rise = height / (ce - lo), fall = height / (hi - ce);
        for (int k = 0; k + 1 < n_bins; ++k) {
            const float f = bin_hz * (float)k;
            if (!(f > lo) || !(f < hi)) continue;
            const float w = f < ce ? (f - lo) * rise : (hi - f) * fall;
            fb[(size_t)m * n_bins + k] = w > 0.0f ? w : 0.0f;
        }
    }
    return fb;
}

// librosa.filters.mel(htk=True, norm=None|'slaney') (export_mfcc_params.py:49-57).
std::vector<float> make_mel_filterbank(int sample_rate, int n_fft, int n_mels, float fmin,
                                       float fmax, int mel_norm)
{
    if (mel_norm == DSP_MELNORM_AUBIO_SLANEY && n_mels == 40) return make_aubio_slaney_filterbank(sample_rate, n_fft);
    const int n_bins = n_fft / 2 + 1;
    // DSP_MELNORM_LIBROSA: librosa.filters.mel's defaults -- Slaney's mel SCALE (htk = False: 200/3 Hz per mel below 1 kHz,
    // log(6.4) / 27 per mel above) with Slaney's area normalisation; what librosa.feature.mfcc, hence cepstrum/train.py:45-52, uses
    const bool slaney_scale = mel_norm == DSP_MELNORM_LIBROSA;
    if (slaney_scale) mel_norm = DSP_MELNORM_SLANEY;
    const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
    auto to_mel = [&](double hz) {
        if (slaney_scale) return hz >= min_log_hz ? min_log_mel + std::log(hz / min_log_hz) / logstep : hz / f_sp;
        return 2595.0 * std::log10(1.0 + hz / 700.0);
    };
    auto to_hz = [&](double mel) {
        if (slaney_scale) return mel >= min_log_mel ? min_log_hz * std::exp(logstep * (mel - min_log_mel)) : f_sp * mel;
        return 700.0 * (std::pow(10.0, mel / 2595.0) - 1.0);
    };
    std::vector<double> edge(n_mels + 2);
    const double lo = to_mel(fmin), hi = to_mel(fmax);
    const double step = (hi - lo) / (n_mels + 1);
    for (int i = 0; i < n_mels + 2; ++i) edge[i] = to_hz(i == n_mels + 1 ? hi : lo + i * step);
    const double bin_hz = 0.5 * sample_rate / (n_bins - 1);
    std::vector<float> fb((size_t)n_mels * n_bins);
    for (int m = 0; m < n_mels; ++m) {
        const double rise = edge[m + 1] - edge[m], fall = edge[m + 2] - edge[m + 1];
        const double area = mel_norm == DSP_MELNORM_SLANEY ? 2.0 / (edge[m + 2] - edge[m]) : 1.0;
        for (int k = 0; k < n_bins; ++k) {
            const double f = (k == n_bins - 1) ? 0.5 * sample_rate : k * bin_hz;
            double w = std::min((f - edge[m]) / rise, (edge[m + 2] - f) / fall);
            if (!(w > 0.0)) w = 0.0;
            float wf = (float)w;
            if (mel_norm == DSP_MELNORM_SLANEY) wf = (float)((double)wf * area);
            fb[(size_t)m * n_bins + k] = wf;
        }
    }
    return fb;
}

// Orthonormal DCT-II rows, float32 argument like the exporter (export_mfcc_params.py:27-41).
std::vector<float> make_dct_ortho(int n_mfcc, int n_mels)
{
    std::vector<float> d((size_t)n_mfcc * n_mels);
    const float pif = (float)kPi;
    const double s0 = std::sqrt(1.0 / n_mels), s1 = std::sqrt(2.0 / n_mels);
    for (int k = 0; k < n_mfcc; ++k)
        for (int m = 0; m < n_mels; ++m) {
            if (k == 0) { d[m] = (float)s0; continue; }
            float arg = pif * ((float)m + 0.5f);
            arg *= (float)k;
            arg /= (float)n_mels;
            d[(size_t)k * n_mels + m] = (float)(s1 * (double)std::cos(arg));
        }
    return d;
}

static void unit(double turns, float &c, float &s)
{
    // exp(-2*pi*i*turns), evaluated in float64
    c = (float)std::cos(-2.0 * kPi * turns);
    s = (float)std::sin(-2.0 * kPi * turns);
}

bool build_gen_tables_2048(const dsp_mfcc_config &cfg, GenTables2048 &t, std::string &why)
{
    std::memset(&t, 0, sizeof(t));
    const int n_fft = 2048, n_bins = 1025;
    if (cfg.n_fft != n_fft) { why = "n_fft must be 2048 for this kernel"; return false; }
    if (cfg.frame_length < 2 || cfg.frame_length > n_fft) { why = "frame_length must be in [2, n_fft]"; return false; }
    if (cfg.n_mels < 1 || cfg.n_mels > k2048MaxMels) { why = "n_mels must be in [1, 128] for n_fft = 2048"; return false; }
    if (cfg.n_mfcc < 1 || cfg.n_mfcc > k2048MaxMfcc) { why = "n_mfcc must be in [1, 32] for n_fft = 2048"; return false; }
    if (cfg.prefilter != DSP_PREFILTER_NONE) { why = "the per-frame prefilter is implemented for n_fft = 512 and 1024"; return false; }
    t.n_mels = cfg.n_mels;
    t.n_mfcc = cfg.n_mfcc;
    std::vector<float> win = make_frame_window(cfg);
    win.resize(n_fft, 0.0f);
    for (int l = 0; l < kLanes; ++l)
        for (int a = 0; a < 16; ++a) {
            const int n = l + 64 * a;
            t.win[2 * a][l] = 0.5f * win[2 * n];
            t.win[2 * a + 1][l] = 0.5f * win[2 * n + 1];
        }
    for (int i = 0; i < 1024; ++i) unit((double)i / 1024.0, t.w1024[0][i], t.w1024[1][i]);
    for (int k = 0; k < 512; ++k) unit((double)k / 2048.0, t.w2048[0][k], t.w2048[1][k]);
    const std::vector<float> fb = make_mel_filterbank(cfg.sample_rate, n_fft, cfg.n_mels, cfg.fmin, cfg.fmax, cfg.mel_norm);
    int off = 0;
    for (int m = 0; m < cfg.n_mels; ++m) {
        const float *row = &fb[(size_t)m * n_bins];
        int first = -1, last = -1;
        for (int k = 0; k < n_bins; ++k)
            if (row[k] != 0.0f) { if (first < 0) first = k; last = k; }
        t.mel_off[m] = off;
        if (first < 0) { t.mel_lo[m] = 0; t.mel_len[m] = 0; continue; }
        const int len = last - first + 1;
        if (off + len > k2048MaxWeights) { why = "mel filterbank has too many non-zero weights for the 2048-point kernel"; return false; }
        t.mel_lo[m] = first;
        t.mel_len[m] = len;
        for (int k = 0; k < len; ++k) t.mel_w[off + k] = row[first + k];       // ascending bins, the reference's summation order (mfcc.c:158-164)
        off += len;
    }
    const std::vector<float> dct = make_dct_ortho(cfg.n_mfcc, cfg.n_mels);
    for (int c = 0; c < cfg.n_mfcc; ++c)
        for (int m = 0; m < cfg.n_mels; ++m) t.dct[c][m] = dct[(size_t)c * cfg.n_mels + m];
    const int half = (cfg.n_mels + 1) / 2;
    for (int l = 0; l < kLanes; ++l) {
        const int c = l >> 1, h = l & 1;
        for (int i = 0; i < half; ++i) {
            const int m = h * half + i;
            t.dct_t[i][l] = (c < cfg.n_mfcc && m < cfg.n_mels) ? t.dct[c][m] : 0.0f;
        }
    }
    // segments: filter m's run [lo, lo + len) in pieces of <= 16 bins, numbered consecutively
    int n_seg = 0;
    bool ok = true;
    for (int m = 0; m < cfg.n_mels && ok; ++m) {
        const int len = t.mel_len[m], pieces = (len + k2048SegTaps - 1) / k2048SegTaps;
        t.mel_s0[m] = n_seg;
        t.mel_cnt[m] = pieces;
        if (pieces > k2048MaxGather || n_seg + pieces > k2048SegSlots * kLanes) { ok = false; break; }
        for (int g = 0; g < pieces; ++g, ++n_seg) {
            const int first = t.mel_lo[m] + g * k2048SegTaps, cnt = std::min(k2048SegTaps, len - g * k2048SegTaps);
            const int k0 = std::min(first, n_bins - k2048SegTaps);            // the 16-bin window stays inside P[0 .. 1024]
            const int slot = n_seg / kLanes, lane = n_seg % kLanes;
            t.seg_k0[slot][lane] = k0;
            for (int j = 0; j < k2048SegTaps; ++j) {
                const int kk = k0 + j;
                t.seg_w[slot][j][lane] = (kk >= first && kk < first + cnt) ? fb[(size_t)m * n_bins + kk] : 0.0f;
            }
        }
    }
    t.seg_ok = ok ? 1 : 0;
    return true;
}

// ---- prefilter in parallel form -------------------------------------------------------------------------------------
bool build_prefilter_scan(const double b[9], const double a[9], PrefilterScan &out, std::string &why)
{
    typedef std::complex<double> cd;
    // poles: roots of z^8 + a1 z^7 + ... + a8 (a0 = 1), Durand-Kerner from points on a spiral, then Newton polish
    cd z[8];
    for (int k = 0; k < 8; ++k) z[k] = std::pow(cd(0.4, 0.9), k);
    auto poly = [&](cd x) { cd v = a[0]; for (int k = 1; k <= 8; ++k) v = v * x + a[k]; return v; };
    for (int it = 0; it < 400; ++it)
        for (int i = 0; i < 8; ++i) {
            cd den = 1.0;
            for (int j = 0; j < 8; ++j) if (j != i) den *= z[i] - z[j];
            z[i] -= poly(z[i]) / den;
        }
    for (int i = 0; i < 8; ++i) {
        if (!(std::abs(poly(z[i])) < 1e-10) || !(std::abs(z[i]) < 1.0)) { why = "prefilter: pole search failed or filter unstable"; return false; }
    }
    out.k0 = b[8] / a[8];
    int ns = 0;
    for (int i = 0; i < 8; ++i) {
        if (!(z[i].imag() > 1e-9)) continue;                       // one of each conjugate pair (Butterworth band-pass: no real poles)
        if (ns == 4) { why = "prefilter: more than four pole pairs"; return false; }
        const cd p = z[i];
        cd B = 0.0, pk = 1.0;                                       // B(1/p)
        for (int k = 0; k <= 8; ++k) { B += b[k] * pk; pk /= p; }
        cd den = 1.0;
        for (int j = 0; j < 8; ++j) if (j != i) den *= 1.0 - z[j] / p;
        const cd r = B / den;                                       // residue of r / (1 - p z^-1)
        out.b0[ns] = 2.0 * r.real();
        out.b1[ns] = -2.0 * (r * std::conj(p)).real();
        out.a1[ns] = -2.0 * p.real();
        out.a2[ns] = std::norm(p);
        {   // smallest D with |p|^(16 * 2^D) < 1e-14
            int D = 0;
            while (D < 6 && std::pow(std::abs(p), 16.0 * (double)(1 << D)) >= 1e-14) ++D;
            out.steps[ns] = D;
        }
        ++ns;
    }
    if (ns != 4) { why = "prefilter: expected four conjugate pole pairs"; return false; }
    for (int sct = 0; sct < 4; ++sct) {
        double m[4] = {-out.a1[sct], -out.a2[sct], 1.0, 0.0}, pwr[4] = {1.0, 0.0, 0.0, 1.0};
        auto mul = [](const double (&x)[4], const double (&y)[4], double (&o)[4]) {
            const double t[4] = {x[0] * y[0] + x[1] * y[2], x[0] * y[1] + x[1] * y[3], x[2] * y[0] + x[3] * y[2], x[2] * y[1] + x[3] * y[3]};
            for (int k = 0; k < 4; ++k) o[k] = t[k];
        };
        for (int k = 0; k < kScanChunk; ++k) mul(pwr, m, pwr);      // M^16
        for (int d = 0; d < 6; ++d) {
            for (int k = 0; k < 4; ++k) out.pw[d][sct][k] = pwr[k];
            mul(pwr, pwr, pwr);
        }
    }
    // self-check against the direct-form-II recurrence (classifier.c:420-446) on 1024 pseudo-random samples
    double x[1024], yref[1024], y[1024], w[8] = {0};
    unsigned long long st = 0x9E3779B97F4A7C15ull;
    for (int n = 0; n < 1024; ++n) { st = st * 6364136223846793005ull + 1442695040888963407ull; x[n] = (double)(st >> 11) / 9007199254740992.0 * 2.0 - 1.0; }
    for (int n = 0; n < 1024; ++n) {
        double w0 = x[n];
        for (int j = 1; j < 9; ++j) w0 -= a[j] * w[j - 1];
        double yy = b[0] * w0;
        for (int j = 1; j < 9; ++j) yy += b[j] * w[j - 1];
        for (int j = 7; j > 0; --j) w[j] = w[j - 1];
        w[0] = w0;
        yref[n] = yy;
    }
    for (int n = 0; n < 1024; ++n) y[n] = out.k0 * x[n];
    for (int sct = 0; sct < 4; ++sct) {
        double T[64][2];
        for (int l = 0; l < 64; ++l) {
            double w1 = 0, w2 = 0;
            for (int i = 0; i < kScanChunk; ++i) { const double w0 = x[16 * l + i] - out.a1[sct] * w1 - out.a2[sct] * w2; w2 = w1; w1 = w0; }
            T[l][0] = w1; T[l][1] = w2;
        }
        for (int d = 0; d < out.steps[sct]; ++d) {
            const double *m = out.pw[d][sct];
            for (int l = 63; l >= (1 << d); --l) {
                const double *u = T[l - (1 << d)];
                T[l][0] += m[0] * u[0] + m[1] * u[1];
                T[l][1] += m[2] * u[0] + m[3] * u[1];
            }
        }
        for (int l = 0; l < 64; ++l) {
            double w1 = l ? T[l - 1][0] : 0.0, w2 = l ? T[l - 1][1] : 0.0;
            for (int i = 0; i < kScanChunk; ++i) {
                const double w0 = x[16 * l + i] - out.a1[sct] * w1 - out.a2[sct] * w2;
                y[16 * l + i] += out.b0[sct] * w0 + out.b1[sct] * w1;
                w2 = w1; w1 = w0;
            }
        }
    }
    double err = 0, top = 0;
    for (int n = 0; n < 1024; ++n) { err = std::max(err, std::fabs(y[n] - yref[n])); top = std::max(top, std::fabs(yref[n])); }
    if (!(err <= 1e-10 * top)) { why = "prefilter: the parallel form does not reproduce the direct form"; return false; }

    // ---- cascade form (see tables.hpp) ---------------------------------------------------------------------------------
    out.c_ok = 0;
    out.c_row_ok = 0;
    {
        const double g = b[0], pat[9] = {1, 0, -4, 0, 6, 0, -4, 0, 1};
        bool binom = g != 0.0;
        for (int k = 0; k < 9 && binom; ++k) binom = std::fabs(b[k] - g * pat[k]) <= 1e-12 * std::fabs(g);
        if (binom) {
            int order[4] = {0, 1, 2, 3};
            std::sort(order, order + 4, [&](int p, int q) { return out.a2[p] < out.a2[q]; });       // a2 = |pole|^2
            out.c_gain = g;
            for (int s = 0; s < 4; ++s) {
                const int src = order[s];
                out.c_a1[s] = out.a1[src]; out.c_a2[s] = out.a2[src];
                out.c_a1f[s] = (float)out.a1[src]; out.c_a2f[s] = (float)out.a2[src];
                const double radius = std::sqrt(out.a2[src]), tol = s < 2 ? 1e-13 : 1e-9;
                int D = 0;
                while (D < 6 && std::pow(radius, 16.0 * (double)(1 << D)) >= tol) ++D;
                out.c_steps[s] = D;
                for (int d = 0; d < 6; ++d)
                    for (int k = 0; k < 4; ++k) { out.c_pw[d][s][k] = out.pw[d][src][k]; out.c_pwf[d][s][k] = (float)out.pw[d][src][k]; }
            }
            // self-check of the cascade's algebra (all four sections in double, the kernel's lane structure) against the direct form
            double u[1024];
            for (int n = 0; n < 1024; ++n) u[n] = x[n];
            for (int s = 0; s < 4; ++s) {
                double T[64][2], v[1024];
                for (int l = 0; l < 64; ++l) {
                    double w1 = 0, w2 = 0;
                    for (int i = 0; i < kScanChunk; ++i) { const double w0 = u[16 * l + i] - out.c_a1[s] * w1 - out.c_a2[s] * w2; w2 = w1; w1 = w0; }
                    T[l][0] = w1; T[l][1] = w2;
                }
                for (int d = 0; d < out.c_steps[s]; ++d) {
                    const double *m = out.c_pw[d][s];
                    for (int l = 63; l >= (1 << d); --l) {
                        const double u0 = T[l - (1 << d)][0], u1 = T[l - (1 << d)][1];
                        T[l][0] += m[0] * u0 + m[1] * u1;
                        T[l][1] += m[2] * u0 + m[3] * u1;
                    }
                }
                for (int l = 0; l < 64; ++l) {
                    double w1 = l ? T[l - 1][0] : 0.0, w2 = l ? T[l - 1][1] : 0.0;
                    for (int i = 0; i < kScanChunk; ++i) {
                        const double w0 = u[16 * l + i] - out.c_a1[s] * w1 - out.c_a2[s] * w2;
                        v[16 * l + i] = w0 - w2;
                        w2 = w1; w1 = w0;
                    }
                }
                for (int n = 0; n < 1024; ++n) u[n] = v[n];
            }
            double cerr = 0;
            for (int n = 0; n < 1024; ++n) cerr = std::max(cerr, std::fabs(g * u[n] - yref[n]));
            out.c_ok = cerr <= 1e-9 * top ? 1 : 0;      // float64 scan steps stop at 1e-13 of the state: 1e-9 leaves room, a wrong table misses by orders

            // ---- row form of the scan: per-lane matrices M^(16 (j + 1)) and the same self-check ------------------------------
            for (int s = 0; s < 4; ++s) {
                double m16[4], pwr[4];
                for (int k = 0; k < 4; ++k) m16[k] = pwr[k] = out.c_pw[0][s][k];                 // M^16
                for (int j = 0; j < 16; ++j) {
                    for (int k = 0; k < 4; ++k) { out.c_rowm[s][j][k] = pwr[k]; out.c_rowmf[s][j][k] = (float)pwr[k]; }
                    const double t[4] = {pwr[0] * m16[0] + pwr[1] * m16[2], pwr[0] * m16[1] + pwr[1] * m16[3],
                                         pwr[2] * m16[0] + pwr[3] * m16[2], pwr[2] * m16[1] + pwr[3] * m16[3]};
                    for (int k = 0; k < 4; ++k) pwr[k] = t[k];
                }
            }
            for (int n = 0; n < 1024; ++n) u[n] = x[n];
            for (int s = 0; s < 4; ++s) {
                double T[64][2], v[1024];
                for (int l = 0; l < 64; ++l) {
                    double w1 = 0, w2 = 0;
                    for (int i = 0; i < kScanChunk; ++i) { const double w0 = u[16 * l + i] - out.c_a1[s] * w1 - out.c_a2[s] * w2; w2 = w1; w1 = w0; }
                    T[l][0] = w1; T[l][1] = w2;
                }
                for (int d = 0; d < scan_row_steps(out.c_steps[s]); ++d) {                         // inside each row
                    const double *m = out.c_pw[d][s];
                    for (int l = 63; l >= 0; --l) {
                        if ((l & 15) < (1 << d)) continue;
                        const double u0 = T[l - (1 << d)][0], u1 = T[l - (1 << d)][1];
                        T[l][0] += m[0] * u0 + m[1] * u1;
                        T[l][1] += m[2] * u0 + m[3] * u1;
                    }
                }
                double W[64][2], C[64][2];
                for (int l = 0; l < 64; ++l) { W[l][0] = C[l][0] = T[l][0]; W[l][1] = C[l][1] = T[l][1]; }
                for (int round = 0; round < scan_row_rounds(out.c_steps[s]); ++round) {           // across rows
                    double N[64][2];
                    for (int l = 0; l < 64; ++l) {
                        const int r = l >> 4, j = l & 15;
                        const double u0 = r ? C[16 * r - 1][0] : 0.0, u1 = r ? C[16 * r - 1][1] : 0.0;
                        const double *m = out.c_rowm[s][j];
                        N[l][0] = W[l][0] + m[0] * u0 + m[1] * u1;
                        N[l][1] = W[l][1] + m[2] * u0 + m[3] * u1;
                    }
                    for (int l = 0; l < 64; ++l) { C[l][0] = N[l][0]; C[l][1] = N[l][1]; }
                }
                for (int l = 0; l < 64; ++l) {
                    double w1 = l ? C[l - 1][0] : 0.0, w2 = l ? C[l - 1][1] : 0.0;
                    for (int i = 0; i < kScanChunk; ++i) {
                        const double w0 = u[16 * l + i] - out.c_a1[s] * w1 - out.c_a2[s] * w2;
                        v[16 * l + i] = w0 - w2;
                        w2 = w1; w1 = w0;
                    }
                }
                for (int n = 0; n < 1024; ++n) u[n] = v[n];
            }
            double rerr = 0;
            for (int n = 0; n < 1024; ++n) rerr = std::max(rerr, std::fabs(g * u[n] - yref[n]));
            out.c_row_ok = rerr <= 1e-9 * top ? 1 : 0;
        }
    }
    return true;
}


bool build_gen_tables_1024(const dsp_mfcc_config &cfg, GenTables1024 &t, std::string &why)
{
    std::memset(&t, 0, sizeof(t));
    const int n_fft = 1024, n_bins = 513;
    if (cfg.n_fft != n_fft) { why = "n_fft must be 1024 for this kernel"; return false; }
    if (cfg.frame_length < 2 || cfg.frame_length > n_fft) { why = "frame_length must be in [2, n_fft]"; return false; }
    if (cfg.n_mels < 1 || cfg.n_mels > kGenMelsPerLane * kLanes) { why = "n_mels must be in [1, 128]"; return false; }
    if (cfg.n_mfcc < 1 || cfg.n_mfcc > 16) { why = "n_mfcc must be in [1, 16] for n_fft = 1024"; return false; }
    t.n_mels = cfg.n_mels;
    t.n_mfcc = cfg.n_mfcc;
    std::vector<float> win = make_frame_window(cfg);
    win.resize(n_fft, 0.0f);
    for (int l = 0; l < kLanes; ++l)
        for (int a = 0; a < 8; ++a) {
            const int n = l + 64 * a;
            t.win[2 * a][l] = 0.5f * win[2 * n];
            t.win[2 * a + 1][l] = 0.5f * win[2 * n + 1];
        }
    for (int l = 0; l < kLanes; ++l)
        for (int i = 0; i < 16; ++i) t.win_chunk[l][i] = 0.5f * win[16 * l + i];
    for (int i = 0; i < 512; ++i) unit((double)i / 512.0, t.w512[0][i], t.w512[1][i]);
    for (int k = 0; k < 256; ++k) unit((double)k / 1024.0, t.w1024[0][k], t.w1024[1][k]);

    // sparse mel: every filter's non-zero run is cut into chunks of <= 12 bins; a chunk goes to one (slot, lane) and is read
    // as a 12-bin window pbuf[k0 .. k0+12).  ds_read_b32 serves lanes 0-31 and 32-63 in separate passes over 32 banks, so
    // the reads of one slot are conflict free iff the windows of each 32-lane half start at distinct addresses mod 32:
    // one bipartite matching (Kuhn) chunk -> (slot, half, start mod 32) over the fewest slots that hold all chunks.
    const std::vector<float> fb = make_mel_filterbank(cfg.sample_rate, n_fft, cfg.n_mels, cfg.fmin, cfg.fmax, cfg.mel_norm);
    for (int i = 0; i < kGenMelsPerLane; ++i)
        for (int g = 0; g < kGenGather; ++g)
            for (int l = 0; l < kLanes; ++l) t.mel_src[i][g][l] = kGenZeroSlot;
    struct Chunk { int m, g, first, len, lo, hi; };
    std::vector<Chunk> chunks;
    for (int m = 0; m < cfg.n_mels; ++m) {
        const float *row = &fb[(size_t)m * n_bins];
        int first = -1, last = -1;
        for (int k = 0; k < n_bins; ++k)
            if (row[k] != 0.0f) { if (first < 0) first = k; last = k; }
        if (first < 0) continue;
        const int run = last - first + 1, pieces = (run + kMelChunk - 1) / kMelChunk;
        if (pieces > kGenGather) { why = "a mel filter spans more than 72 bins"; return false; }
        int k = first;
        for (int g = 0; g < pieces; ++g) {
            const int len = run / pieces + (g < run % pieces ? 1 : 0);
            // the window [k0, k0+12) must cover the chunk and stay inside [0, 512]
            chunks.push_back({m, g, k, len, std::max(0, k + len - kMelChunk), std::min(k, n_bins - kMelChunk)});
            k += len;
        }
    }
    const int nc = (int)chunks.size();
    if (nc > kGenChunks * kLanes) { why = "mel filterbank needs more than 256 chunks of 12 bins"; return false; }
    const int n_slots = std::max(1, (nc + kLanes - 1) / kLanes);
    const int n_pos = n_slots * 64;                          // position = (slot * 2 + half) * 32 + start mod 32
    std::vector<int> owner(n_pos, -1), pos(nc, -1), k0(nc, -1);
    std::vector<char> seen;
    struct Rec { static bool go(int i, const std::vector<Chunk> &c, int n_groups, std::vector<int> &owner, std::vector<char> &seen,
                                std::vector<int> &pos, std::vector<int> &k0) {
        for (int g = 0; g < n_groups; ++g)
            for (int k = c[i].hi; k >= c[i].lo; --k) {
                const int p = 32 * g + (k & 31);
                if (seen[p]) continue;
                seen[p] = 1;
                if (owner[p] < 0 || go(owner[p], c, n_groups, owner, seen, pos, k0)) {
                    owner[p] = i; pos[i] = p; k0[i] = k;
                    return true;
                }
            }
        return false; } };
    bool placed = true;
    for (int i = 0; i < nc && placed; ++i) {
        seen.assign(n_pos, 0);
        placed = Rec::go(i, chunks, 2 * n_slots, owner, seen, pos, k0);
    }
    std::vector<int> used(2 * n_slots, 0);
    int next = 0;
    for (int i = 0; i < nc; ++i) {
        const Chunk &c = chunks[i];
        int slot, lane;
        if (placed) {
            const int g = pos[i] / 32;
            slot = g / 2;
            lane = 32 * (g % 2) + used[g]++;
        } else {                                             // still correct, just not conflict free
            slot = next / kLanes; lane = next % kLanes; k0[i] = c.hi;
        }
        ++next;
        const float *row = &fb[(size_t)c.m * n_bins];
        t.mel_k0[slot][lane] = k0[i];
        for (int j = 0; j < kMelChunk; ++j) {
            const int kk = k0[i] + j;
            t.mel_w[slot][j][lane] = (kk >= c.first && kk < c.first + c.len) ? row[kk] : 0.0f;
        }
        t.mel_src[c.m / kLanes][c.g][c.m % kLanes] = slot * kLanes + lane;     // partial index
    }
    t.n_chunk_slots = n_slots;
    // idle lanes (weights all 0) still issue the 12 reads: leave their windows at 0 (broadcast-free but harmless)
    const std::vector<float> dct = make_dct_ortho(cfg.n_mfcc, cfg.n_mels);
    for (int c = 0; c < cfg.n_mfcc; ++c)
        for (int q = 0; q < 4; ++q)
            for (int i = 0; i < kGenDctLen; ++i) {
                const int m = q * kGenDctLen + i;
                t.dct_w[i][4 * c + q] = m < cfg.n_mels ? dct[(size_t)c * cfg.n_mels + m] : 0.0f;
            }
    for (int st = 0; st < kGenDctSteps; ++st)
        for (int l = 0; l < kLanes; ++l) {
            const int c = l % 16, m = 4 * st + l / 16;
            t.dct_a[st][l] = (c < cfg.n_mfcc && m < cfg.n_mels) ? dct[(size_t)c * cfg.n_mels + m] : 0.0f;
        }
    for (int l = 0; l < kLanes; ++l) {
        for (int q = 1; q < 8; ++q) {
            unit((double)(l * q) / 512.0, t.tw1[2 * (q - 1)][l], t.tw1[2 * (q - 1) + 1][l]);
            unit((double)((l % 8) * q) / 64.0, t.tw2[2 * (q - 1)][l], t.tw2[2 * (q - 1) + 1][l]);
        }
        for (int tt = 0; tt < 4; ++tt) unit((double)(l + 64 * tt) / 1024.0, t.twp[2 * tt][l], t.twp[2 * tt + 1][l]);
    }
    return true;
}

void build_pair_extra_512(PairExtra512 &t)
{
    std::memset(&t, 0, sizeof(t));
    for (int l = 0; l < kLanes; ++l) {
        const int f = (l >> 2) & 1, j = (l & 3) + 4 * (l >> 3);
        for (int p = 1; p < 8; ++p) unit((double)((l % 8) * p) / 64.0, t.tw2[2 * (p - 1)][l], t.tw2[2 * (p - 1) + 1][l]);
        for (int tt = 0; tt < 4; ++tt) unit((double)(j + 32 * tt) / 512.0, t.twp[2 * tt][l], t.twp[2 * tt + 1][l]);
        const int jp = (32 - j) & 31;
        t.partner[l] = (jp & 3) | (f << 2) | ((jp >> 2) << 3);
    }
}

void build_row_tables_512(const dsp_mfcc_config &cfg, RowTables512 &t)
{
    std::memset(&t, 0, sizeof(t));
    std::vector<float> win = make_frame_window(cfg);
    win.resize(512, 0.0f);
    for (int l = 0; l < kLanes; ++l) {
        const int j = l & 15;
        for (int k = 0; k < 16; ++k) {
            const int n = j + 16 * k;
            t.win[2 * k][l] = 0.5f * win[2 * n];          // x0.5: see build_lane_tables_512
            t.win[2 * k + 1][l] = 0.5f * win[2 * n + 1];
        }
        for (int q = 1; q <= 15; ++q) unit((double)(j * q) / 256.0, t.tw[2 * (q - 1)][l], t.tw[2 * (q - 1) + 1][l]);
        for (int m = 0; m < 8; ++m) unit((double)(j + 16 * m) / 512.0, t.twp[2 * m][l], t.twp[2 * m + 1][l]);
    }
}

bool build_lane_tables_512(const dsp_mfcc_config &cfg, LaneTables512 &t, std::string &why)
{
    std::memset(&t, 0, sizeof(t));
    const int n_fft = 512, n_bins = 257;
    if (cfg.n_fft != n_fft) { why = "n_fft must be 512 for this kernel"; return false; }
    if (cfg.frame_length < 2 || cfg.frame_length > n_fft) { why = "frame_length must be in [2, n_fft]"; return false; }
    if (cfg.n_mels < 1 || cfg.n_mels > kLanes) { why = "n_mels must be in [1, 64]"; return false; }
    if (cfg.n_mfcc < 1 || cfg.n_mfcc > 32) { why = "n_mfcc must be in [1, 32]"; return false; }
    t.n_mels = cfg.n_mels;
    t.n_mfcc = cfg.n_mfcc;

    // window, pre-scaled by 1/2: the packed real FFT untangling X[k] =
    // ((Z[k]+Z*[N/2-k]) - i W^k (Z[k]-Z*[N/2-k]))/2 then needs no scaling
    // (multiplying by 0.5 is exact in binary floating point).
    std::vector<float> win = make_frame_window(cfg);
    win.resize(n_fft, 0.0f);
    for (int l = 0; l < kLanes; ++l)
        for (int a = 0; a < 4; ++a) {
            const int n = l + 64 * a;
            t.win[2 * a][l] = 0.5f * win[2 * n];
            t.win[2 * a + 1][l] = 0.5f * win[2 * n + 1];
        }
    for (int l = 0; l < kLanes; ++l) {
        for (int q = 1; q <= 3; ++q) {
            unit((double)(l * q) / 256.0, t.tw1[2 * (q - 1)][l], t.tw1[2 * (q - 1) + 1][l]);
            unit((double)((l & 15) * q) / 64.0, t.tw2[2 * (q - 1)][l], t.tw2[2 * (q - 1) + 1][l]);
            unit((double)((l & 3) * q) / 16.0, t.tw3[2 * (q - 1)][l], t.tw3[2 * (q - 1) + 1][l]);
        }
    }
    // After the FFT lane l holds bins 64 t + kappa(l).  With exchange 3 through LDS
    // the reader lanes are chosen so that kappa = l; with the register-only (DPP)
    // exchange lane l = (beta, p, o) (base-4 digits) ends with kappa = 16 o + 4 p + beta.  The
    // conjugate partner of bin k is bin 256 - k, i.e. kappa' = (64 - kappa) mod 64.
    int lane_of[kLanes];
    for (int l = 0; l < kLanes; ++l) {
        t.kappa[l] = DSP_X3_LDS ? l : 16 * (l & 3) + 4 * ((l >> 2) & 3) + (l >> 4);
        lane_of[t.kappa[l]] = l;
    }
    for (int l = 0; l < kLanes; ++l) {
        t.partner[l] = lane_of[(64 - t.kappa[l]) & 63];
        unit((double)t.kappa[l] / 512.0, t.twp[0][l], t.twp[1][l]);
        unit((double)(t.kappa[l] + 64) / 512.0, t.twp[2][l], t.twp[3][l]);
    }

    // sparse mel: every filter's non-zero run is cut into chunks of <= 12 bins,
    // one chunk per lane (494 non-zeros in 61 chunks for the reference config).
    const std::vector<float> fb = make_mel_filterbank(cfg.sample_rate, n_fft, cfg.n_mels, cfg.fmin,
                                                      cfg.fmax, cfg.mel_norm);
    for (int g = 0; g < kMelGather; ++g)
        for (int l = 0; l < kLanes; ++l) t.mel_src[g][l] = kZeroSlot;
    struct Chunk { int m, g, first, len, lo, hi; };
    std::vector<Chunk> chunks;
    for (int m = 0; m < cfg.n_mels; ++m) {
        const float *row = &fb[(size_t)m * n_bins];
        int first = -1, last = -1;
        for (int k = 0; k < n_bins; ++k)
            if (row[k] != 0.0f) { if (first < 0) first = k; last = k; }
        if (first < 0) continue;  // empty filter -> energy 0 (all sources = zero slot)
        // cut the run into the fewest chunks of <= 12 bins, lengths as even as possible:
        // shorter-than-12 chunks leave their 12-bin read window room to slide
        const int run = last - first + 1;
        const int pieces = (run + kMelChunk - 1) / kMelChunk;
        if (pieces > kMelGather) { why = "a mel filter spans more than 72 bins"; return false; }
        int k = first;
        for (int g = 0; g < pieces; ++g) {
            const int len = run / pieces + (g < run % pieces ? 1 : 0);
            // the window [k0, k0+12) must cover the chunk and stay inside [0, 256]
            const int lo = std::max(0, k + len - kMelChunk), hi = std::min(k, n_bins - kMelChunk);
            chunks.push_back({m, g, k, len, lo, hi});
            if (g + 1 > t.mel_gather) t.mel_gather = g + 1;
            k += len;
        }
    }
    t.mel_gather = t.mel_gather <= 3 ? 3 : 6;
    if ((int)chunks.size() > kLanes) { why = "mel filterbank needs more than 64 chunks of 12 bins"; return false; }

    // Placement: ds_read_b32 serves lanes 0-31 and 32-63 in separate passes over 32
    // banks, so the 12 window reads are conflict free iff the windows of each half
    // start at distinct addresses mod 32.  One bipartite matching (Kuhn): chunk ->
    // slot (half, start mod 32), 64 slots.
    const int nc = (int)chunks.size();
    std::vector<int> half(nc, 0), k0(nc, -1);
    int owner[64];
    for (int &o : owner) o = -1;
    std::vector<char> seen;
    struct Rec { static bool go(int i, const std::vector<Chunk> &c, int *owner, std::vector<char> &seen,
                                std::vector<int> &half, std::vector<int> &k0) {
        for (int h = 0; h < 2; ++h)
            for (int k = c[i].lo; k <= c[i].hi; ++k) {
                const int slot = 32 * h + (k & 31);
                if (seen[slot]) continue;
                seen[slot] = 1;
                if (owner[slot] < 0 || go(owner[slot], c, owner, seen, half, k0)) {
                    owner[slot] = i; half[i] = h; k0[i] = k;
                    return true;
                }
            }
        return false; } };
    bool placed = true;
    for (int i = 0; i < nc && placed; ++i) {
        seen.assign(64, 0);
        placed = Rec::go(i, chunks, owner, seen, half, k0);
    }
    if (!placed) {   // still correct, just not conflict free
        for (int i = 0; i < nc; ++i) { half[i] = i < 32 ? 0 : 1; k0[i] = chunks[i].hi; }
    }
    int used[2] = {0, 0};
    for (int i = 0; i < nc; ++i) {
        const Chunk &c = chunks[i];
        const int lane = 32 * half[i] + used[half[i]]++;
        const float *row = &fb[(size_t)c.m * n_bins];
        t.mel_k0[lane] = k0[i];
        for (int j = 0; j < kMelChunk; ++j) {
            const int kk = k0[i] + j;
            t.mel_w[j][lane] = (kk >= c.first && kk < c.first + c.len) ? row[kk] : 0.0f;
        }
        t.mel_src[c.g][c.m] = lane;
    }
    t.mel_conflict_free = placed ? 1 : 0;
    // idle lanes (weights all 0) still issue the 12 reads: park them on residues
    // no live window of their half uses
    for (int h = 0; h < 2; ++h) {
        bool taken[32] = {false};
        for (int l = 0; l < used[h]; ++l) taken[t.mel_k0[32 * h + l] & 31] = true;
        int r = 0;
        for (int l = used[h]; l < 32; ++l) {
            while (r < 32 && taken[r]) ++r;
            t.mel_k0[32 * h + l] = r < 32 ? r : 0;
            if (r < 32) taken[r] = true;
        }
    }

    // DCT: split the n_mels-long dot product over 4 (or 2) neighbouring lanes.
    t.dct_split = cfg.n_mfcc <= 16 ? 4 : 2;
    // the kernel is instantiated for (split, len) in {(4,10), (4,16), (2,20)}:
    // take the smallest instantiated length that covers n_mels (extra weights are 0)
    const int need = (cfg.n_mels + t.dct_split - 1) / t.dct_split;
    int len;
    if (t.dct_split == 4 && need <= 10) len = 10;
    else if (t.dct_split == 4 && need <= 16) len = 16;
    else if (t.dct_split == 2 && need <= 20) len = 20;
    else { why = "n_mels too large for this n_mfcc (need n_mels <= 64 for n_mfcc <= 16, <= 40 otherwise)"; return false; }
    t.dct_len = len;
    const std::vector<float> dct = make_dct_ortho(cfg.n_mfcc, cfg.n_mels);
    for (int c = 0; c < cfg.n_mfcc; ++c)
        for (int q = 0; q < t.dct_split; ++q) {
            const int lane = t.dct_split * c + q;
            for (int i = 0; i < len; ++i) {
                const int m = q * len + i;
                t.dct_w[i][lane] = m < cfg.n_mels ? dct[(size_t)c * cfg.n_mels + m] : 0.0f;
            }
        }
    for (int ct = 0; ct < 2; ++ct)
        for (int s = 0; s < kDctSteps; ++s)
            for (int l = 0; l < kLanes; ++l) {
                const int c = 16 * ct + l % 16, m = 4 * s + l / 16;
                t.dct_a[ct][s][l] = (c < cfg.n_mfcc && m < cfg.n_mels) ? dct[(size_t)c * cfg.n_mels + m] : 0.0f;
            }
    return true;
}

}  // namespace dsp
