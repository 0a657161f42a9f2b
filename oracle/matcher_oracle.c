/* matcher_oracle.c -- CPU ORACLE of cv::BFMatcher(NORM_HAMMING)::knnMatch(k = 2) at full size (test infrastructure only; never
 * imported, linked or called by the product).  The same published algorithm oracle/matcher_oracle.py restates -- Hamming distance
 * = popcount(xor), the train descriptors scanned in order, a candidate replaces a kept one only on a strictly smaller distance,
 * so ties go to the lower train index (OpenCV's batch_distance + sort of (distance, index) pairs gives the same order) -- as
 * plain C with OpenMP over the queries and 64-bit popcounts, so that 40 000 x 40 000 x 512 bit (1.6e9 descriptor pairs) takes
 * seconds instead of the numpy form's hours.  Call sites in the reference: samples/sample_image_sequence.cpp:114-144,
 * sample_feature_matching.cpp:99-101.  parity unpinned (OpenCV is not in this image; no golden vectors exist). */
#include <stdint.h>
#include <string.h>

/* idx / dist: nq x 2 int32, -1 where fewer than two trains exist.  desc_bytes: a multiple of 8 */
void efxo_knn2_hamming(const uint8_t* query, int nq, const uint8_t* train, int nt, int desc_bytes, int32_t* idx, int32_t* dist)
{
    const int nw = desc_bytes / 8;
#pragma omp parallel for schedule(dynamic, 64)
    for (int i = 0; i < nq; i++) {
        uint64_t q[16];
        memcpy(q, query + (size_t)i * desc_bytes, (size_t)desc_bytes);
        int b0 = -1, b1 = -1, d0 = 1 << 30, d1 = 1 << 30;
        for (int j = 0; j < nt; j++) {
            uint64_t t[16];
            memcpy(t, train + (size_t)j * desc_bytes, (size_t)desc_bytes);
            int d = 0;
            for (int w = 0; w < nw; w++) d += __builtin_popcountll(q[w] ^ t[w]);
            if (d < d0) { d1 = d0; b1 = b0; d0 = d; b0 = j; }
            else if (d < d1) { d1 = d; b1 = j; }
        }
        idx[2 * i] = b0; idx[2 * i + 1] = b1;
        dist[2 * i] = b0 >= 0 ? d0 : -1; dist[2 * i + 1] = b1 >= 0 ? d1 : -1;
    }
}
