// Input generator of SURVEY.md Appendix A (the driver the survey stage used to run the compiled
// reference).  Reproduces the exact (meta, x, y, theta) the recorded reference nlml values of
// SURVEY.md section 8c belong to.  libstdc++-specific (std::*_distribution), which is why its OUTPUT
// is committed as a fixture (appendixA_*.bin) and this program is only the provenance record.
//   usage: gen D N Q R variant out.bin
//   variant 0: persistent distribution objects; 1: fresh temporaries per draw
// File layout (little endian): int32 D,N,Q,R,H | int32 meta[N] | float x[N] | float y[N] | double theta[H]
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
int main(int argc, char **argv) {
    if (argc != 7) return 2;
    int D = atoi(argv[1]), N = atoi(argv[2]), Q = atoi(argv[3]), R = atoi(argv[4]), variant = atoi(argv[5]);
    std::mt19937 g(1234);
    std::vector<int32_t> meta(N);
    std::vector<float> x(N), y(N);
    std::uniform_real_distribution<float> ux(0, 200);
    std::normal_distribution<float> ny(0, 1);
    for (int i = 0; i < N; i++) {
        meta[i] = i % D;
        if (variant == 0) { x[i] = ux(g); y[i] = ny(g); }
        else { x[i] = std::uniform_real_distribution<float>(0, 200)(g); y[i] = std::normal_distribution<float>(0, 1)(g); }
    }
    auto U = [&](double a, double b) { return std::uniform_real_distribution<double>(a, b)(g); };
    std::vector<double> th;
    for (int d = 0; d < D; d++) th.push_back(std::log(U(0.15, 0.4)));
    for (int i = 0; i < Q * D * R; i++) th.push_back(U(-1.5, 1.5) * 0.9 / std::sqrt((double)(Q * R)));
    for (int q = 0; q < Q; q++) th.push_back(std::log(1.0 / U(12, 72)));
    for (int q = 0; q < Q; q++) th.push_back(std::log(1.0 / (2 * 3.14159265 * U(6, 72))));
    for (int i = 0; i < Q * D; i++) th.push_back(std::log(U(0.1, 0.5) * 0.1 / Q));
    int32_t H = (int32_t)th.size();
    FILE *f = fopen(argv[6], "wb");
    int32_t hdr[5] = {D, N, Q, R, H};
    fwrite(hdr, 4, 5, f); fwrite(meta.data(), 4, N, f); fwrite(x.data(), 4, N, f); fwrite(y.data(), 4, N, f);
    fwrite(th.data(), 8, H, f); fclose(f);
    return 0;
}
