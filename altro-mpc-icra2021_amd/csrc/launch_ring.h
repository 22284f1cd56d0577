// launch_ring.h -- launch-duration history of a handle: a FIXED ring of HIP event pairs.
// One pair is recorded around every solve launch (the measurement behind altro_batch_timing_get and
// bench.py's roofline figure).  The ring reuses its events, so an MPC consumer that calls
// altro_mpc_step_async once per tick for hours creates no new runtime objects; only the most
// recent CAP launches can be read back.
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

namespace altro {

struct LaunchRing {
  static constexpr size_t CAP = 1024;
  std::vector<hipEvent_t> ev;  // up to 2 * CAP events, created on first use of a slot
  size_t count = 0;            // launches recorded since the last reset

  // events of the next launch (start, end); creates them the first time the slot is used
  hipError_t next(hipEvent_t* start, hipEvent_t* end) {
    const size_t slot = count % CAP;
    while (ev.size() < 2 * (slot + 1)) {
      hipEvent_t e;
      const hipError_t rc = hipEventCreate(&e);
      if (rc != hipSuccess) return rc;
      ev.push_back(e);  // capacity is reserved in reset(): cannot throw here
    }
    *start = ev[2 * slot];
    *end = ev[2 * slot + 1];
    count++;
    return hipSuccess;
  }
  void reset() {
    if (ev.capacity() < 2 * CAP) ev.reserve(2 * CAP);
    count = 0;
  }
  size_t readable() const { return count < CAP ? count : CAP; }
  // i-th oldest readable launch (0 = oldest still held)
  hipError_t elapsed(size_t i, float* ms) const {
    const size_t first = count - readable();
    const size_t slot = (first + i) % CAP;
    return hipEventElapsedTime(ms, ev[2 * slot], ev[2 * slot + 1]);
  }
  void destroy() {
    for (hipEvent_t e : ev) hipEventDestroy(e);
    ev.clear();
    count = 0;
  }
};

}  // namespace altro
