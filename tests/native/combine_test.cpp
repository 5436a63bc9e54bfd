// Host-only test of csrc/combine.hpp (the combining queue behind anr_encoder_forward_shared) with a stand-in forward:
// out[b][h] = (sum of the row's first lens[b] ids) * (h + 1) * (normalize ? 0.5 : 1) + (types ? 1000 : 0).
// Built and run by tests/test_encoder_combiner_cpu.py (g++ -pthread).  Prints one JSON line.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

#include "combine.hpp"

using anr::ForwardCombiner;

static std::atomic<int> g_inside{0}, g_max_inside{0}, g_calls{0}, g_max_rows{0}, g_max_tokens{0};
static constexpr int kHidden = 8;

static std::atomic<int> g_lane_inside[8];
static std::atomic<int> g_lane_clash{0};
static int fake_forward(int lane, const int32_t *ids, const int32_t *lens, const int32_t *types, int B, int L, int norm, float *out,
                        std::string *err) {
  if (++g_lane_inside[lane] != 1) ++g_lane_clash;  // a lane runs one forward at a time
  const int in = ++g_inside;
  int m = g_max_inside.load();
  while (in > m && !g_max_inside.compare_exchange_weak(m, in)) {}
  ++g_calls;
  int mr = g_max_rows.load();
  while (B > mr && !g_max_rows.compare_exchange_weak(mr, B)) {}
  const int tok = B * ((L + 31) / 32 * 32);
  int mt = g_max_tokens.load();
  while (tok > mt && !g_max_tokens.compare_exchange_weak(mt, tok)) {}
  std::this_thread::sleep_for(std::chrono::microseconds(300));
  int rc = 0;
  for (int b = 0; b < B; ++b) {
    long s = 0;
    for (int i = 0; i < lens[b]; ++i) s += ids[(size_t)b * L + i];
    if (ids[(size_t)b * L] == 666) {  // a poisoned request: the forward that holds it fails
      rc = -2;
      *err = "poisoned forward";
    }
    for (int h = 0; h < kHidden; ++h) out[(size_t)b * kHidden + h] = (float)(s * (h + 1)) * (norm ? 0.5f : 1.f) + (types ? 1000.f : 0.f);
  }
  --g_inside;
  --g_lane_inside[lane];
  return rc;
}

int main(int argc, char **argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 8, N = argc > 2 ? atoi(argv[2]) : 200, LANES = argc > 3 ? atoi(argv[3]) : 1;
  ForwardCombiner comb(LANES);
  std::atomic<long> wrong{0}, failed{0}, poisoned_ok{0};
  auto worker = [&](int w) {
    unsigned seed = 1234u + (unsigned)w;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
    for (int i = 0; i < N; ++i) {
      const int B = 1 + (int)(rnd() % 3), L = 5 + (int)(rnd() % 90);
      const bool typed = (w % 4) == 3, norm = (w % 2) == 0;  // threads differ in flags: only compatible requests may merge
      const bool poison = (w == 1 && i % 37 == 5);
      std::vector<int32_t> ids((size_t)B * L), lens(B), types((size_t)B * L, 0);
      for (int b = 0; b < B; ++b) {
        lens[b] = 1 + (int)(rnd() % L);
        for (int j = 0; j < L; ++j) ids[(size_t)b * L + j] = j < lens[b] ? 1 + (int)(rnd() % 500) : 7777;  // padding must not count
      }
      if (poison) ids[0] = 666;
      std::vector<float> out((size_t)B * kHidden, -1.f);
      ForwardCombiner::Req req{ids.data(), lens.data(), typed ? types.data() : nullptr, B, L, norm ? 1 : 0, out.data()};
      const int rc = comb.run(req, kHidden, fake_forward);
      if (rc != 0) {
        ++failed;
        if (req.err == "poisoned forward") ++poisoned_ok;
        continue;
      }
      for (int b = 0; b < B; ++b) {
        long s = 0;
        for (int j = 0; j < lens[b]; ++j) s += ids[(size_t)b * L + j];
        for (int h = 0; h < kHidden; ++h) {
          const float want = (float)(s * (h + 1)) * (norm ? 0.5f : 1.f) + (typed ? 1000.f : 0.f);
          if (out[(size_t)b * kHidden + h] != want) ++wrong;
        }
      }
    }
  };
  std::vector<std::thread> th;
  const auto t0 = std::chrono::steady_clock::now();
  for (int w = 0; w < T; ++w) th.emplace_back(worker, w);
  for (auto &t : th) t.join();
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  int64_t fw = 0, rq = 0;
  comb.stats(&fw, &rq);
  const int storm_rows = g_max_rows.load(), storm_tokens = g_max_tokens.load();
  // a big request alone, after the storm: the queue must be idle and usable again
  std::vector<int32_t> ids(64 * 40, 3), lens(64, 40);
  std::vector<float> out(64 * kHidden);
  ForwardCombiner::Req big{ids.data(), lens.data(), nullptr, 64, 40, 0, out.data()};
  const int rc_big = comb.run(big, kHidden, fake_forward);
  printf("{\"lanes\": %d, \"lane_clash\": %d, \"threads\": %d, \"requests\": %ld, \"served\": %ld, \"forwards\": %ld, \"calls\": %d, \"wrong\": %ld, \"failed\": %ld, "
         "\"poisoned_ok\": %ld, \"max_concurrent_forwards\": %d, \"max_rows\": %d, \"max_tokens\": %d, \"ms\": %.1f, \"rc_big\": %d, "
         "\"big0\": %.1f}\n",
         LANES, g_lane_clash.load(), T, (long)T * N, (long)rq, (long)fw, g_calls.load(), wrong.load(), failed.load(), poisoned_ok.load(), g_max_inside.load(),
         storm_rows, storm_tokens, ms, rc_big, out[0]);
  return 0;
}
