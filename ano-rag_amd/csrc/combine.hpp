// Combining queue for small forwards (anr_encoder_forward_shared).  Host-only C++ (no HIP): tests/native/combine_test.cpp
// drives it with a stand-in forward under g++.
//
// The reference answers questions from worker threads that share ONE model and call it with a single query each
// (main_musique.py:487-494, query/query_processor.py:2761-2766).  Serialised at the encoder's mutex that is T x 0.5 ms of
// device time back to back, while one forward of eight queries costs little more than one of one.  A combining queue in
// Python (round 4, first half) reached 3.5x at 8 threads and no further: between two forwards the leading thread had to win
// the interpreter lock back from the other callers — 0.1-0.2 ms of every 0.8 ms cycle with the device idle.  This queue lives
// below the interpreter: every caller is inside the C call with the interpreter lock released.  Whoever finds no forward
// running leads: it takes the oldest request and everything queued that may share its forward — same normalize flag, same use
// of token types, at most kMaxTokens padded tokens, so that the merged forward stays in the kernel regime of a single query
// and no embedding depends on who else was in flight (token rows are independent in every kernel, and a key block that lies
// wholly in a sequence's padding adds exact zeros to its attention sums: bit-identical to the one-at-a-time call) — runs ONE
// forward, scatters the rows to the callers' buffers and wakes them.  The lead serves until its own request is done and three
// rounds have passed (waking a new leader costs more than a round), then hands over to the oldest waiter.
// LANES: a forward of a few queries is ~90 dependent launches of tiny kernels — latency, not work — so two of them side by
// side on two streams take hardly longer than one.  The queue therefore admits up to `lanes` leaders at a time, each bound
// to a lane index that the forward callback maps to its own stream and workspace (anr_encoder: the handle itself and a view
// of it that shares the weights).  A request that has been given the lead is skipped by the other leaders' rounds.
#pragma once
#include <algorithm>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <deque>
#include <exception>
#include <mutex>
#include <string>
#include <vector>

namespace anr {

class ForwardCombiner {
 public:
  static constexpr int kMaxTokens = 2048;
  static constexpr int kLeadRounds = 3;
  struct Req {
    const int32_t *ids, *lens, *types;  // [B][L], [B], [B][L] or nullptr
    int B, L, normalize;
    float *out;                         // [B][hidden]
    int rc = 0;
    std::string err;
    bool done = false, lead = false;
    int lane = -1;
  };

  explicit ForwardCombiner(int lanes = 1) : lanes_(lanes < 1 ? 1 : (lanes > 8 ? 8 : lanes)) {}

  // fwd(lane, ids, lens, types, B, L, normalize, out, &err) -> 0 or an error code; blocks until req is served
  template <typename F>
  int run(Req &req, int hidden, F &&fwd) {
    std::unique_lock<std::mutex> lk(mu_);
    queue_.push_back(&req);
    for (int l = 0; l < lanes_; ++l)
      if (!(busy_ & (1u << l))) {
        busy_ |= 1u << l;
        req.lead = true;
        req.lane = l;
        break;
      }
    cv_.wait(lk, [&] { return req.done || req.lead; });
    if (!req.lead) return req.rc;
    // the lead of lane req.lane; this caller's own request is still queued (the other leaders skip it)
    const int lane = req.lane;
    std::vector<Req *> part;
    std::vector<int32_t> ids, lens, types;
    std::vector<float> out;
    for (int rounds = 1;; ++rounds) {
      part.clear();
      int rows = 0, Lp = 0, Lmax = 0, first_norm = 0;
      bool typed = false;
      for (auto it = queue_.begin(); it != queue_.end();) {
        Req *r = *it;
        if (r->lead && r != &req) {  // about to lead another lane's round: its own round serves it
          ++it;
          continue;
        }
        if (part.empty()) {
          first_norm = r->normalize;
          typed = r->types != nullptr;
        }
        const int Lq = std::max(Lp, (r->L + 31) / 32 * 32);
        const bool same = r->normalize == first_norm && (r->types != nullptr) == typed;
        if (same && (part.empty() || (int64_t)(rows + r->B) * Lq <= kMaxTokens)) {
          part.push_back(r);
          rows += r->B;
          Lp = Lq;
          Lmax = std::max(Lmax, r->L);
          it = queue_.erase(it);
        } else {
          ++it;
        }
      }
      if (part.empty()) {  // (own request done, and what is queued belongs to the other lanes' leaders)
        busy_ &= ~(1u << lane);
        break;
      }
      lk.unlock();
      int rc = 0;
      std::string msg;
      try {
        if (part.size() == 1) {
          Req *r = part[0];
          rc = fwd(lane, r->ids, r->lens, r->types, r->B, r->L, r->normalize, r->out, &msg);
        } else {
          ids.assign((size_t)rows * Lmax, 0);  // padding: any valid token (masked by the lengths)
          if (typed) types.assign((size_t)rows * Lmax, 0);
          lens.resize(rows);
          out.resize((size_t)rows * hidden);
          int a = 0;
          for (Req *r : part) {
            for (int b = 0; b < r->B; ++b) {
              std::memcpy(ids.data() + (size_t)(a + b) * Lmax, r->ids + (size_t)b * r->L, (size_t)r->L * sizeof(int32_t));
              if (typed)
                std::memcpy(types.data() + (size_t)(a + b) * Lmax, r->types + (size_t)b * r->L, (size_t)r->L * sizeof(int32_t));
              lens[a + b] = r->lens[b];
            }
            a += r->B;
          }
          rc = fwd(lane, ids.data(), lens.data(), typed ? types.data() : nullptr, rows, Lmax, first_norm, out.data(), &msg);
          a = 0;
          if (rc == 0)
            for (Req *r : part) {
              std::memcpy(r->out, out.data() + (size_t)a * hidden, (size_t)r->B * hidden * sizeof(float));
              a += r->B;
            }
        }
      } catch (const std::exception &ex) {  // (allocation failure while merging: the round's callers get the error, the lane lives on)
        rc = -4;
        msg = ex.what();
      }
      lk.lock();
      forwards_ += 1;
      served_ += (int64_t)part.size();
      for (Req *r : part) {
        r->rc = rc;
        r->err = msg;
        r->done = true;
      }
      cv_.notify_all();
      if (req.done && (queue_.empty() || rounds >= kLeadRounds)) {
        // leave: the lane goes to the oldest waiter that is not leading already, or is free again
        Req *next = nullptr;
        for (Req *r : queue_)
          if (!r->lead) {
            next = r;
            break;
          }
        if (next) {
          next->lead = true;
          next->lane = lane;
          cv_.notify_all();
        } else {
          busy_ &= ~(1u << lane);
        }
        break;
      }
    }
    return req.rc;
  }

  void stats(int64_t *forwards, int64_t *requests) {
    std::lock_guard<std::mutex> lk(mu_);
    if (forwards) *forwards = forwards_;
    if (requests) *requests = served_;
  }

 private:
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<Req *> queue_;
  const int lanes_;
  unsigned busy_ = 0;  // lanes that have a leader
  int64_t forwards_ = 0, served_ = 0;
};

}  // namespace anr
