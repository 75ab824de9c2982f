// TAA jitter sequence shared by the host library (blok_taa_jitter) and the backend (blok_hip_draw_frame_rt).
// Reference: PostProcess::initJitterSequence / halton / advanceJitter, blok/src/renderer_postprocess.cpp:208-228,243,660-663.
#ifndef BLOK_TAA_JITTER_H
#define BLOK_TAA_JITTER_H
#include <stdint.h>

namespace blok {

// PostProcess::halton (:216-228): the same float operations in the same order.
inline float halton(int index, int base) {
    float result = 0.0f;
    float f = 1.0f / static_cast<float>(base);
    int i = index;
    while (i > 0) {
        result += f * static_cast<float>(i % base);
        i = i / base;
        f = f / static_cast<float>(base);
    }
    return result;
}

// Entry i of the sequence = (halton(i + 1, 2) - 0.5, halton(i + 1, 3) - 0.5), 16 entries (:208-214); the index starts at 0
// and advances once per frame (swapHistoryBuffers -> advanceJitter), so frame k uses entry k mod 16.
inline void taa_jitter_px(uint32_t frame_index, float out_px[2]) {
    const int i = static_cast<int>(frame_index % 16u);
    out_px[0] = halton(i + 1, 2) - 0.5f;
    out_px[1] = halton(i + 1, 3) - 0.5f;
}

}  // namespace blok
#endif
