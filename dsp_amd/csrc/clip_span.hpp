// clip_span.hpp -- one clip of a ragged batch (the reference's callers loop over files of different lengths --
// cepstrum/scrubjay_infer.c:158-176, 2fa/audio/word/c/main_test.c:254-331, donut-classifier/classifier.c:286-297): where it starts in the
// input buffer, its samples, and the frames (MFCC frames, or spectrogram segments for the classifiers) the host counted for it -- the
// kernels divide nothing.  16 bytes: one scalar load per clip.
#pragma once

namespace dsp {

struct ClipSpan {
    long off;      // first sample (per channel) from the start of the input buffer
    int n;         // samples per channel
    int frames;    // MFCC frames (>= 1) / spectrogram segments (>= 0)
};

}  // namespace dsp
