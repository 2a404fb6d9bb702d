/*
 * main_files.c -- a plain-C caller shaped like the main() of donut-classifier/classifier.c (a loop over audio files of any
 * length, :286-297 reads channel 0 of each): every file of the command line is read as 16-bit PCM, all of them go to the GPU in
 * ONE ragged call -- dsp_classify_batch_ragged_pcm16_host, the float32 firmware arithmetic of sync/lib/classifier.cpp, or with
 * -d the float64 classifier of donut-classifier/classifier.c -- and the program prints the reference's line per file.
 *
 *   gcc -O2 -Iinclude examples/main_files.c -Ldsp_amd -ldsp_amd -Wl,-rpath,$PWD/dsp_amd -o main_files
 *   ./main_files [-d] a.wav b.wav ...            (all files mono or all stereo)
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dsp_amd.h"

/* appends the file's samples (interleaved as stored) to *buf; returns sample frames read, < 0 on error */
static long read_wav_pcm16(const char *path, int16_t **buf, long *used, long *cap, int *channels)
{
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    uint8_t hdr[12];
    if (fread(hdr, 1, 12, f) != 12 || memcmp(hdr, "RIFF", 4) || memcmp(hdr + 8, "WAVE", 4)) { fclose(f); return -1; }
    int ch = 1, bits = 16;
    long frames = -1;
    for (;;) {
        uint8_t ck[8];
        if (fread(ck, 1, 8, f) != 8) break;
        const uint32_t size = ck[4] | ck[5] << 8 | ck[6] << 16 | (uint32_t)ck[7] << 24;
        if (!memcmp(ck, "fmt ", 4)) {
            uint8_t fmt[16];
            if (size < 16 || fread(fmt, 1, 16, f) != 16) break;
            ch = fmt[2] | fmt[3] << 8;
            bits = fmt[14] | fmt[15] << 8;
            fseek(f, (long)size - 16, SEEK_CUR);
        } else if (!memcmp(ck, "data", 4)) {
            if (bits != 16 || (ch != 1 && ch != 2) || (*channels && *channels != ch)) break;
            *channels = ch;
            const long n = (long)size / 2;                      /* int16 values */
            if (*used + n > *cap) {
                *cap = 2 * (*used + n);
                *buf = (int16_t *)realloc(*buf, (size_t)*cap * sizeof(int16_t));
                if (!*buf) break;
            }
            if ((long)fread(*buf + *used, 2, (size_t)n, f) != n) break;
            *used += n;
            frames = n / ch;
            break;
        } else {
            fseek(f, (long)size + (size & 1), SEEK_CUR);
        }
    }
    fclose(f);
    return frames;
}

int main(int argc, char **argv)
{
    int first = 1, f64 = 0;
    if (argc > 1 && !strcmp(argv[1], "-d")) { f64 = 1; first = 2; }
    const int n_files = argc - first;
    if (n_files <= 0) { fprintf(stderr, "usage: %s [-d] file.wav ...\n", argv[0]); return 2; }
    int16_t *pcm = NULL;
    long used = 0, cap = 0;
    int channels = 0;
    long *offsets = (long *)calloc((size_t)n_files + 1, sizeof(long));
    int *labels = (int *)calloc((size_t)n_files, sizeof(int));
    for (int i = 0; i < n_files; ++i) {
        const long frames = read_wav_pcm16(argv[first + i], &pcm, &used, &cap, &channels);
        if (frames < 0) { fprintf(stderr, "%s: not a 16-bit PCM WAV (or its channel count differs from the files before it)\n", argv[first + i]); return 1; }
        offsets[i + 1] = offsets[i] + frames;                   /* sample frames per channel */
    }
    /* one call for all files: clip i = frames [offsets[i], offsets[i + 1]) of the buffer, channel 0 as classifier.c:292-297 */
    const int rc = f64 ? dsp_classify_batch_ragged_pcm16_host_f64(NULL, pcm, n_files, offsets, channels, DSP_STEREO_CHANNEL0, labels, NULL)
                       : dsp_classify_batch_ragged_pcm16_host(NULL, pcm, n_files, offsets, channels, DSP_STEREO_CHANNEL0, labels, NULL);
    if (rc < 0) { fprintf(stderr, "libdsp_amd: %s\n", dsp_last_error()); return 1; }
    for (int i = 0; i < n_files; ++i)
        printf(labels[i] ? "%s has a Scrub Jay! :)\n" : "%s has no Scrub Jay! :(\n", argv[first + i]);      /* classifier.c:186-191 */
    free(pcm); free(offsets); free(labels);
    return 0;
}
