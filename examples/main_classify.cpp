// main_classify.cpp -- a C++ caller of the donut classifier exactly as sync/sync.cpp:202 is one: it includes the
// reference's own classifier.h (or, where the reference tree is absent, include/dsp_amd_classifier.h, which repeats
// those declarations) and links libdsp_amd.so INSTEAD of classifier.cpp + PlainFFT.cpp:
//
//   g++ -DUSE_REFERENCE_HEADER -I$REF/sync/lib examples/main_classify.cpp -Ldsp_amd -ldsp_amd -o main_classify
//   g++ -Iinclude                             examples/main_classify.cpp -Ldsp_amd -ldsp_amd -o main_classify
//
//   main_classify clip.f32 out.bin [map.f32 nf nt lower upper half_range midpoint freqs.f32 times.f32]
//
// clip.f32: raw float32 mono PCM at 16 kHz.  out.bin (for tests/test_boundary_cxx.py), all 4-byte little endian:
//   ok(butter 3000-7500) b[9] a[9] ok(bad band) n filtered[n] freq_bins time_bins freqs[] times[] Sxx[freq_bins][time_bins]
//   n_midpoints midpoints[] label [sum_intense]
#ifdef USE_REFERENCE_HEADER
#include "classifier.h"
#else
#include "dsp_amd_classifier.h"
#endif

#include <cstdio>
#include <cstdlib>
#include <vector>

static std::vector<float> read_f32(const char *path)
{
    std::vector<float> v;
    FILE *f = std::fopen(path, "rb");
    if (!f) { std::perror(path); std::exit(2); }
    float buf[4096];
    size_t got;
    while ((got = std::fread(buf, sizeof(float), 4096, f)) > 0) v.insert(v.end(), buf, buf + got);
    std::fclose(f);
    return v;
}

int main(int argc, char **argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: %s clip.f32 out.bin [map.f32 nf nt lower upper half mid freqs.f32 times.f32]\n", argv[0]); return 2; }
    std::vector<float> x = read_f32(argv[1]);
    const int n = (int)x.size();
    FILE *o = std::fopen(argv[2], "wb");
    if (!o) { std::perror(argv[2]); return 2; }
    auto put_i = [&](int v) { std::fwrite(&v, 4, 1, o); };
    auto put_f = [&](const float *p, size_t k) { std::fwrite(p, 4, k, o); };

    float b[9] = {0}, a[9] = {0}, b2[9], a2[9];
    put_i(butter_bandpass(3000.0f, 7500.0f, b, a) ? 1 : 0);
    put_f(b, 9); put_f(a, 9);
    put_i(butter_bandpass(2000.0f, 6000.0f, b2, a2) ? 1 : 0);

    std::vector<float> y(n);
    butter_bandpass_filter(x.data(), n, b, a, y.data());
    put_i(n); put_f(y.data(), n);

    float *freqs = nullptr, *times = nullptr, **sxx = nullptr;
    int nf = 0, nt = 0;
    compute_spectrogram(y.data(), n, 16000, &freqs, &times, &sxx, &nf, &nt);
    put_i(nf); put_i(nt); put_f(freqs, nf); put_f(times, nt);
    for (int i = 0; i < nf; ++i) { put_f(sxx[i], nt); std::free(sxx[i]); }       // the caller frees (classifier.cpp:126-133)
    std::free(sxx); std::free(freqs); std::free(times);

    int n_mid = 0;
    float *mids = find_midpoints(x.data(), n, 16000, &n_mid);
    put_i(n_mid); put_f(mids, n_mid);
    std::free(mids);

    put_i(classify(x.data(), n));

    if (argc >= 12) {
        std::vector<float> map = read_f32(argv[3]);
        const int mf = std::atoi(argv[4]), mt = std::atoi(argv[5]);
        std::vector<float *> rows(mf);
        for (int i = 0; i < mf; ++i) rows[i] = map.data() + (size_t)i * mt;
        std::vector<float> fr = read_f32(argv[10]), tm = read_f32(argv[11]);
        const float s = sum_intense((float)std::atof(argv[6]), (float)std::atof(argv[7]), (float)std::atof(argv[8]), fr.data(), mf, tm.data(), mt,
                                    rows.data(), (float)std::atof(argv[9]));
        put_f(&s, 1);
    }
    std::fclose(o);
    std::printf("n=%d T=%d midpoints=%d\n", n, nt, n_mid);
    return 0;
}
