// asan_driver.cpp — the host emulation of the wave programs (factorizer_amd/csrc/nmf_core.h, nmf_gram.h through tests/emul/emul.cpp)
// run under AddressSanitizer + UndefinedBehaviorSanitizer (SURVEY.md §5 "sanitizers": GPU ASan is not available on this pool, so the
// programs are sanitized where they compile for the host): out-of-bounds indices into the LDS-history images, the factor arrays
// and the masked tails of ragged matrices would abort here.  Built and run by tests/test_wave_program_emul.py::test_emulation_under_asan.
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

extern "C" int emu_gram_bwd(const float* x, const float* v0, const float* gy, float* gx, int64_t nmat, int M, int N, int T, int G,
                            float eps, float gscale);
extern "C" int emu_nmf_fwd(const float* x, const float* u0, const float* v0, float* y, float* uo, float* vo, int64_t nmat, int M,
                           int N, int R, int T, int solver, float eps);
extern "C" int emu_nmf_bwd(const float* x, const float* u0, const float* v0, const float* gy, const float* gu, const float* gv,
                           float* gx, int64_t nmat, int M, int N, int R, int T, int G, int solver, float eps);

int main() {
  std::mt19937 rng(7);
  std::uniform_real_distribution<float> U(0.f, 1.f);
  struct Case { int M, N, R, T, G, solver; };
  const Case cases[] = {{8, 512, 1, 5, 5, 1}, {8, 512, 2, 5, 5, 0}, {8, 150, 2, 10, 10, 1}, {8, 64, 3, 5, 2, 1}, {5, 37, 2, 4, 4, 0},
                        {8, 200, 1, 3, 1, 1}, {8, 256, 4, 3, 3, 0}, {1, 1, 1, 2, 2, 1}};
  int bad = 0;
  for (const Case& c : cases) {
    const int64_t nmat = 3;
    // exact-size buffers: one element past any of them is a sanitizer report
    std::vector<float> x(nmat * c.M * c.N), gy(x.size()), y(x.size()), gx(x.size()), u0(c.M * c.R), v0(c.N * c.R),
        uo(nmat * c.M * c.R), vo(nmat * c.N * c.R);
    for (auto& v : x) v = U(rng);
    for (auto& v : gy) v = U(rng) - 0.5f;
    for (auto& v : u0) v = U(rng);
    for (auto& v : v0) v = U(rng);
    for (int k = 0; k < c.M * c.N; ++k) x[k] = 0.f;   // an all-zero matrix (eps paths)
    int rc = emu_nmf_fwd(x.data(), u0.data(), v0.data(), y.data(), uo.data(), vo.data(), nmat, c.M, c.N, c.R, c.T, c.solver, 1e-16f);
    if (rc != 0) { std::printf("fwd rc %d for %dx%d R%d\n", rc, c.M, c.N, c.R); ++bad; }
    rc = emu_nmf_bwd(x.data(), u0.data(), v0.data(), gy.data(), nullptr, nullptr, gx.data(), nmat, c.M, c.N, c.R, c.T, c.G, c.solver, 1e-16f);
    if (rc != 0) { std::printf("bwd rc %d for %dx%d R%d\n", rc, c.M, c.N, c.R); ++bad; }
    rc = emu_nmf_bwd(x.data(), u0.data(), v0.data(), nullptr, uo.data(), vo.data(), gx.data(), nmat, c.M, c.N, c.R, c.T, c.G, c.solver, 1e-16f);
    if (rc != 0) { std::printf("bwd(gu, gv) rc %d for %dx%d R%d\n", rc, c.M, c.N, c.R); ++bad; }
    for (float v : gx) if (!(v == v)) { std::printf("NaN in gx for %dx%d R%d\n", c.M, c.N, c.R); ++bad; break; }
    if (c.R == 1 && c.solver == 1 && c.M <= 8 && c.N <= 512) {   // the row-space backward (HALS rank 1 on non-negative input)
      rc = emu_gram_bwd(x.data(), v0.data(), gy.data(), gx.data(), nmat, c.M, c.N, c.T, c.G, 1e-16f, 2.0f);
      if (rc != 0) { std::printf("gram rc %d for %dx%d\n", rc, c.M, c.N); ++bad; }
    }
  }
  std::printf("asan driver: %d problem(s)\n", bad);
  return bad ? 1 : 0;
}
