// Section timers of the rebvio API: REBVIO_TIMER_TICK / TOCK and their NAMED forms (reference util/timer.hpp:18-32).
// Compiled in with -DTIMER like the reference; otherwise the macros expand to nothing. A function-static Timer
// accumulates the time between tick() and tock() and prints one summary when it is destroyed at exit. On this backend
// a tick/tock pair around a hot-path call times the ENQUEUE of its kernels unless the call synchronises; device-side
// stage times come from rebvio_hip_profile_* (bench.py reports them at the same sites).
#pragma once

#include <chrono>
#include <cstdio>
#include <string>

#ifdef TIMER
#define REBVIO_TIMER_TICK()                                    \
  static rebvio::util::Timer timer(__PRETTY_FUNCTION__);       \
  timer.tick();
#define REBVIO_NAMED_TIMER_TICK(NAME)                          \
  static rebvio::util::Timer timer_##NAME(#NAME);              \
  timer_##NAME.tick();
#define REBVIO_TIMER_TOCK() timer.tock();
#define REBVIO_NAMED_TIMER_TOCK(NAME) timer_##NAME.tock();
#else
#define REBVIO_TIMER_TICK()
#define REBVIO_TIMER_TOCK()
#define REBVIO_NAMED_TIMER_TICK(NAME)
#define REBVIO_NAMED_TIMER_TOCK(NAME)
#endif

namespace rebvio {
namespace util {

class Timer {
 public:
  explicit Timer(std::string name) : name_(std::move(name)) {}
  ~Timer() { report(); }

  void tick() { start_ = clock::now(); }
  void tock() {
    total_us_ += std::chrono::duration<double, std::micro>(clock::now() - start_).count();
    ++passes_;
  }
  long passes() const { return passes_; }
  double total_ms() const { return total_us_ * 1e-3; }

 private:
  using clock = std::chrono::steady_clock;
  void report() const {
    std::printf("\n[rebvio timer] %s: %ld passes, %.3f ms total, %.4f ms average\n", name_.c_str(), passes_, total_ms(),
                passes_ ? total_ms() / (double)passes_ : 0.0);
  }
  std::string name_;
  clock::time_point start_{};
  double total_us_ = 0.0;
  long passes_ = 0;
};

}  // namespace util
}  // namespace rebvio
