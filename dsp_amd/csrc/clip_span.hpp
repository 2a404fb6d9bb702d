// clip_span.hpp -- one clip of a ragged batch (the reference's callers loop over files of different lengths --
// cepstrum/scrubjay_infer.c:158-176, 2fa/audio/word/c/main_test.c:254-331, donut-classifier/classifier.c:286-297): where it starts in the
// input buffer, its samples, and the frames (MFCC frames, or spectrogram segments for the classifiers) the host counted for it -- the
// kernels divide nothing -- and, for the fused clip kernels, the caller's index of the clip: the host lays the spans out in the order that
// balances the kernels' fixed deal of clips to wavefronts, the results go to `orig`.  32 bytes: one scalar load per clip.
#pragma once

namespace dsp {

struct ClipSpan {
    long off;      // first sample (per channel) from the start of the input buffer
    int n;         // samples per channel
    int frames;    // MFCC frames (>= 1) / spectrogram segments (>= 0)
    long orig;     // the clip's index in the caller's batch (where its results go)
    long reserved;
};
static_assert(sizeof(ClipSpan) == 32, "one aligned scalar load");

}  // namespace dsp
