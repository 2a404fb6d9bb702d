/*
 * main_mfcc.c -- a plain-C caller of the drop-in entry point, shaped like the
 * reference's host test main (2fa/audio/word/c/main_test.c:254-331): read a
 * 16-bit PCM WAV, convert to float, call compute_mfcc(), print the first frame.
 *
 *   gcc -O2 -Iinclude examples/main_mfcc.c -Ldsp_amd -ldsp_amd -Wl,-rpath,$PWD/dsp_amd -o main_mfcc
 *   ./main_mfcc clip.wav
 *
 * The only change against a build that links the reference's mfcc.c is the
 * library on the link line; the call site is identical.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dsp_amd.h"

#define MAX_FRAMES 500 /* stop_detector.c:9 */
#define N_MFCC 13

static float *read_wav_mono(const char *path, int *n_out)
{
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    uint8_t hdr[12];
    if (fread(hdr, 1, 12, f) != 12 || memcmp(hdr, "RIFF", 4) || memcmp(hdr + 8, "WAVE", 4)) { fclose(f); return NULL; }
    int channels = 1, bits = 16;
    float *sig = NULL;
    for (;;) {
        uint8_t ck[8];
        if (fread(ck, 1, 8, f) != 8) break;
        uint32_t size = ck[4] | ck[5] << 8 | ck[6] << 16 | (uint32_t)ck[7] << 24;
        if (!memcmp(ck, "fmt ", 4)) {
            uint8_t fmt[16];
            if (fread(fmt, 1, 16, f) != 16) break;
            channels = fmt[2] | fmt[3] << 8;
            bits = fmt[14] | fmt[15] << 8;
            fseek(f, (long)size - 16, SEEK_CUR);
        } else if (!memcmp(ck, "data", 4)) {
            if (bits != 16 || channels < 1) break;
            int frames = (int)(size / 2 / channels);
            int16_t *pcm = (int16_t *)malloc(size);
            if (fread(pcm, 1, size, f) != size) { free(pcm); break; }
            sig = (float *)malloc(sizeof(float) * frames);
            for (int i = 0; i < frames; ++i) {
                /* stereo -> average, main_test.c:205-217 */
                float acc = 0.0f;
                for (int c = 0; c < channels; ++c) acc += pcm[i * channels + c] / 32768.0f;
                sig[i] = acc / channels;
            }
            free(pcm);
            *n_out = frames;
            break;
        } else {
            fseek(f, (long)size + (size & 1), SEEK_CUR);
        }
    }
    fclose(f);
    return sig;
}

int main(int argc, char **argv)
{
    int n = 16000;
    float *signal;
    if (argc > 1) {
        signal = read_wav_mono(argv[1], &n);
        if (!signal) { fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
    } else { /* no file: one second of a 440 Hz square-ish test tone */
        signal = (float *)malloc(sizeof(float) * n);
        for (int i = 0; i < n; ++i) signal[i] = ((i / 18) & 1) ? 0.25f : -0.25f;
    }
    static float mfcc[MAX_FRAMES * N_MFCC];
    int frames = compute_mfcc(signal, n, mfcc, MAX_FRAMES); /* same call as stop_detector.c:18 */
    if (frames == 0 && n >= 400) fprintf(stderr, "compute_mfcc: %s\n", dsp_last_error());
    printf("%d samples -> %d frames\n", n, frames);
    for (int c = 0; c < N_MFCC && frames > 0; ++c) printf("%s%.4f", c ? " " : "frame 0: ", mfcc[c]);
    if (frames > 0) printf("\n");
    free(signal);
    return frames > 0 || n < 400 ? 0 : 2;
}
