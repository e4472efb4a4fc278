// host_batch.hpp — the stop protocol of a multi-process solve (no HIP in here: tests/cpp/
// test_batch_protocol.cpp drives it with a simulated device).
//
// Every rank must queue the SAME number of solver iterations: each iteration of a column shard
// holds a collective (the all-gather of the pass's block), and a rank that stopped queueing
// one iteration earlier than its peers would leave them waiting in it forever. No rank may
// therefore decide to stop on anything but data that is bit-identical on all ranks: the
// SolveShared record (every rank recomputes all O(m) work on the same gathered bits). The loop
// queues iterations in batches; after batch n it queues a snapshot of the record (an async
// copy + an event) and, BEFORE looking at it, queues batch n + 1 — only then does it wait for
// snapshot n. The device is never idle, and all ranks read `done` from the same snapshot index,
// hence stop after the same batch. Iterations queued past convergence are no-ops on the device
// but still perform their exchange (the kernels exit on `done`, the collective does not know).
#pragma once

#include <cstdint>

namespace clipper_hip {

// enqueue()            queue one solver iteration (incl. its exchange); 0 or an error code
// snapshot(slot)       queue a copy of the shared record into host slot `slot` (0 | 1); 0 or error
// wait_done(slot, d)   wait for that copy, d = its `done` flag; 0 or error
// Returns 0 or the first error; *iterations = how many iterations were queued.
template <class Enqueue, class Snapshot, class WaitDone>
int run_batched_until_done(int batch, Enqueue&& enqueue, Snapshot&& snapshot, WaitDone&& wait_done,
                           int64_t* iterations) {
  int slot = 0;
  bool have_prev = false, done = false;
  int64_t n = 0;
  if (batch < 1) batch = 1;
  while (!done) {
    for (int it = 0; it < batch; ++it) {
      if (int rc = enqueue()) return rc;
      ++n;
    }
    if (int rc = snapshot(slot)) return rc;
    if (have_prev) {
      bool d = false;
      if (int rc = wait_done(slot ^ 1, d)) return rc;
      done = d;
    }
    have_prev = true;
    slot ^= 1;
  }
  if (iterations) *iterations = n;
  return 0;
}

}  // namespace clipper_hip
