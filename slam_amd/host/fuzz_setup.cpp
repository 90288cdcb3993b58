// Fuzzes the host-only entry points of the C ABI (slam_hip_ellipse_setup, slam_hip_project_setup: float and double
// arithmetic ending in integer conversions) with special values, random bit patterns and plain numbers, and checks
// the invariants of what they report.  Meant for a sanitizer build of the library's host code (no GPU needed):
//   for f in context pyramid match fast newpoints project; do
//     hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I include -I slam_amd/csrc \
//           -Xarch_host -fsanitize=undefined -c slam_amd/csrc/$f.hip -o /tmp/ub/$f.o; done
//   hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=undefined -o /tmp/ub/libslam_hip_ubsan.so /tmp/ub/*.o
//   clang++ -O1 -g -std=c++17 -I include -fsanitize=undefined -fno-sanitize-recover=undefined \
//           slam_amd/host/fuzz_setup.cpp -o /tmp/ub/fuzz -L/tmp/ub -lslam_hip_ubsan -Wl,-rpath,/tmp/ub
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <limits>
#include "slam_hip.h"
static float fuzz(unsigned& s) {
    s = s * 1664525u + 1013904223u;
    const float specials[] = {0.f, -0.f, 1.f, -1.f, 1e-30f, 1e30f, -1e30f, std::numeric_limits<float>::infinity(),
                              -std::numeric_limits<float>::infinity(), std::numeric_limits<float>::quiet_NaN(), 3.4e38f, 1e9f, -1e9f, 0.5f, 2147483648.f};
    if ((s >> 28) < 5) return specials[(s >> 8) % (sizeof specials / sizeof specials[0])];
    unsigned bits = s ^ (s << 13);
    float f; memcpy(&f, &bits, 4);
    return ((s >> 27) & 1) ? f : (float)((int)(s >> 10) % 2000 - 1000) / 7.f;
}
int main() {
    unsigned s = 12345;
    long valid = 0;
    for (int it = 0; it < 2000000; it++) {
        float cov[3] = {fuzz(s), fuzz(s), fuzz(s)};
        slam_ellipse_job e;
        slam_hip_ellipse_setup(fuzz(s), fuzz(s), cov, fuzz(s), fuzz(s), &e);
        valid += e.valid;
        if (e.valid && (e.w < 1 || e.h < 1 || e.w > 300 || e.h > 300)) { printf("bad window\n"); return 1; }
        float P[12], B[12], p3[3], m[9]; int32_t loc[4], ok;
        for (float& v : P) v = fuzz(s);
        for (float& v : B) v = fuzz(s);
        for (float& v : p3) v = fuzz(s);
        slam_hip_project_setup(P, B, p3, (int)(s % 90), (int)((s >> 8) % 90), loc, m, &ok);
        valid += ok;
        if (ok && (loc[2] < 1 || loc[3] < 1 || loc[2] > 500 || loc[3] > 500)) { printf("bad location %d %d\n", loc[2], loc[3]); return 1; }
    }
    printf("fuzzed, %ld valid results\n", valid);
    printf("scratch %zu %zu\n", slam_hip_fast_scratch_bytes(1920, 1080, 64, 65536), slam_hip_fast_scratch_bytes(-1, 5, 1, 1));
    printf("null ctx: %d %d\n", slam_hip_synchronize(nullptr), slam_hip_track_batch(nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr));
    return 0;
}
