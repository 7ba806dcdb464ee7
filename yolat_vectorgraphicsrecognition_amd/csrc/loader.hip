// loader.hip — the batch hand-over off the consumer's thread (round 5).
//
// Reference: cad_recognition/train.py:178-189 hides collate behind DataLoader(num_workers=8) worker PROCESSES, and
// architecture3cc_rpn_gp_iter2.py:107-115 then copies six tensors to the GPU synchronously inside forward().  Here ONE
// native worker thread (no GIL, no Python) per loader
//   * waits for a free slot of a small ring (pinned staging buffer + device buffer),
//   * runs yolat_collate_batch (layout + memcpy of every key + merge of the items' cached CSRs) into the pinned buffer,
//   * enqueues ONE asynchronous host -> device copy on the loader's own HIP stream and records an event,
// while the consumer is still enqueueing / running the forward of the previous batch.  yolat_loader_next makes the
// consumer's stream wait on that event (a stream-side wait, the host does not block on the copy) and hands out the
// device buffer with the field offsets; yolat_loader_release records an event on the consumer's stream after which the
// slot's buffers may be rewritten.  Nothing here computes: the bytes are those of yolat_collate_batch (bit-exact tests
// in tests/test_gpu_loader.py).
#include "common.hpp"
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

namespace {
struct Slot {
  void* pin = nullptr; size_t pin_cap = 0;
  void* dev = nullptr; size_t dev_cap = 0;
  hipEvent_t copied = nullptr, released = nullptr;
  bool release_pending = false;      // `released` has been recorded and not yet waited for
  bool in_use = false;               // handed to the consumer (between next and release) or being filled
  // description of the batch the slot holds
  int64_t off[YOLAT_MAX_KEYS + 6], total = 0, totals[3] = {0, 0, 0}, n_keys = 0, B = 0;
  std::vector<int64_t> slices;
  int rc = 0;
};
struct Request { std::vector<const yolat_item_desc*> items; };
}  // namespace

struct yolat_loader {
  int device = 0, nslots = 0;
  hipStream_t copy_stream = nullptr;
  std::vector<Slot> slots;
  std::mutex mu;
  std::condition_variable cv_req, cv_ready, cv_free;
  std::deque<Request> requests;      // submitted, not yet picked up by the worker
  std::deque<int> ready;             // slots whose copy has been enqueued, in submission order
  long submitted = 0, produced = 0;
  bool stop = false;
  std::thread worker;
};

static void loader_main(yolat_loader* L) {
  (void)hipSetDevice(L->device);
  int next_slot = 0;
  for (;;) {
    Request rq;
    {
      std::unique_lock<std::mutex> lk(L->mu);
      L->cv_req.wait(lk, [&] { return L->stop || !L->requests.empty(); });
      if (L->stop) return;
      rq = std::move(L->requests.front());
      L->requests.pop_front();
    }
    const int si = next_slot;
    next_slot = (next_slot + 1) % L->nslots;
    Slot& s = L->slots[(size_t)si];
    bool wait_release = false;
    {
      std::unique_lock<std::mutex> lk(L->mu);
      L->cv_free.wait(lk, [&] { return L->stop || !s.in_use; });
      if (L->stop) return;
      s.in_use = true;
      wait_release = s.release_pending;
      s.release_pending = false;
    }
    // the consumer's kernels that read the slot's device buffer are done (and with them the copy that filled it)
    if (wait_release) (void)hipEventSynchronize(s.released);
    const int64_t B = (int64_t)rq.items.size();
    s.B = B;
    s.n_keys = B > 0 && rq.items[0] ? rq.items[0]->n_keys : 0;
    s.rc = 0;
    if (s.n_keys < 0 || s.n_keys > YOLAT_MAX_KEYS) s.rc = YOLAT_E_INVALID;
    if (s.rc == 0) {
      s.slices.assign((size_t)(s.n_keys * (B + 1) + 1), 0);
      // layout pass (no destination), then make room, then the real pass
      s.rc = yolat_collate_batch(rq.items.data(), B, nullptr, 0, s.off, &s.total, s.slices.data(), s.totals);
    }
    if (s.rc == 0 && (size_t)s.total > s.pin_cap) {
      if (s.pin) (void)hipHostFree(s.pin);
      s.pin = nullptr;
      s.pin_cap = (size_t)s.total + (size_t)s.total / 2;
      if (hipHostMalloc(&s.pin, s.pin_cap, hipHostMallocDefault) != hipSuccess) { s.pin_cap = 0; s.rc = YOLAT_E_INVALID; }
    }
    if (s.rc == 0 && (size_t)s.total > s.dev_cap) {
      if (s.dev) (void)hipFree(s.dev);
      s.dev = nullptr;
      s.dev_cap = (size_t)s.total + (size_t)s.total / 2;
      if (hipMalloc(&s.dev, s.dev_cap) != hipSuccess) { s.dev_cap = 0; s.rc = YOLAT_E_INVALID; }
    }
    if (s.rc == 0)
      s.rc = yolat_collate_batch(rq.items.data(), B, s.pin, (int64_t)s.pin_cap, s.off, &s.total, s.slices.data(), s.totals);
    if (s.rc == 0) {
      if (hipMemcpyAsync(s.dev, s.pin, (size_t)s.total, hipMemcpyHostToDevice, L->copy_stream) != hipSuccess ||
          hipEventRecord(s.copied, L->copy_stream) != hipSuccess)
        s.rc = YOLAT_E_INVALID;
    }
    {
      std::lock_guard<std::mutex> lk(L->mu);
      L->ready.push_back(si);
      ++L->produced;
    }
    L->cv_ready.notify_all();
  }
}

extern "C" yolat_loader* yolat_loader_create(int slots) {
  if (slots < 2 || slots > 16) return nullptr;
  yolat_loader* L = new yolat_loader();
  if (hipGetDevice(&L->device) != hipSuccess) { delete L; return nullptr; }
  L->nslots = slots;
  L->slots.resize((size_t)slots);
  bool ok = hipStreamCreateWithFlags(&L->copy_stream, hipStreamNonBlocking) == hipSuccess;
  for (Slot& s : L->slots) {
    ok = ok && hipEventCreateWithFlags(&s.copied, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&s.released, hipEventDisableTiming) == hipSuccess;
  }
  if (!ok) {
    for (Slot& s : L->slots) { if (s.copied) (void)hipEventDestroy(s.copied); if (s.released) (void)hipEventDestroy(s.released); }
    if (L->copy_stream) (void)hipStreamDestroy(L->copy_stream);
    delete L;
    return nullptr;
  }
  L->worker = std::thread(loader_main, L);
  return L;
}

extern "C" int yolat_loader_submit(yolat_loader* L, const yolat_item_desc* const* items, int64_t B) {
  if (!L || !items || B <= 0) return YOLAT_E_INVALID;
  Request rq;
  rq.items.assign(items, items + B);
  {
    std::lock_guard<std::mutex> lk(L->mu);
    L->requests.push_back(std::move(rq));
    ++L->submitted;
  }
  L->cv_req.notify_one();
  return 0;
}

extern "C" int yolat_loader_next(yolat_loader* L, yolat_stream_t consumer_stream, yolat_loader_batch* out) {
  if (!L || !out) return YOLAT_E_INVALID;
  int si;
  {
    std::unique_lock<std::mutex> lk(L->mu);
    if (L->ready.empty() && L->requests.empty() && L->produced == L->submitted) return YOLAT_E_INVALID;   // nothing in flight
    L->cv_ready.wait(lk, [&] { return !L->ready.empty(); });
    si = L->ready.front();
    L->ready.pop_front();
  }
  Slot& s = L->slots[(size_t)si];
  out->slot = si;
  out->rc = s.rc;
  if (s.rc != 0) {                                 // the slot goes straight back
    std::lock_guard<std::mutex> lk(L->mu);
    s.in_use = false;
    L->cv_free.notify_all();
    return s.rc;
  }
  // The worker runs a batch or two ahead, so the copy has usually landed by now: work enqueued after a COMPLETED event sees
  // the copy without a stream-side wait, and the consumer's queue is spared a barrier packet on another queue's signal
  // (measured at cfg 2: that wait left the GPU idle for ~14 us of every 113 us hand-over).
  if (hipEventQuery(s.copied) != hipSuccess) {
    (void)hipGetLastError();                       // hipErrorNotReady is not an error here
    if (hipStreamWaitEvent((hipStream_t)consumer_stream, s.copied, 0) != hipSuccess) return YOLAT_E_INVALID;
  }
  out->device = s.dev;
  out->total = s.total;
  out->n_keys = s.n_keys;
  out->B = s.B;
  for (int i = 0; i < YOLAT_MAX_KEYS + 6; ++i) out->off[i] = s.off[i];
  out->N = s.totals[0]; out->E = s.totals[1]; out->P = s.totals[2];
  out->slices = s.slices.data();
  return 0;
}

extern "C" int yolat_loader_release(yolat_loader* L, int slot, yolat_stream_t consumer_stream) {
  if (!L || slot < 0 || slot >= L->nslots) return YOLAT_E_INVALID;
  Slot& s = L->slots[(size_t)slot];
  if (hipEventRecord(s.released, (hipStream_t)consumer_stream) != hipSuccess) return YOLAT_E_INVALID;
  {
    std::lock_guard<std::mutex> lk(L->mu);
    s.release_pending = true;
    s.in_use = false;
  }
  L->cv_free.notify_all();
  return 0;
}

extern "C" void yolat_loader_destroy(yolat_loader* L) {
  if (!L) return;
  {
    std::lock_guard<std::mutex> lk(L->mu);
    L->stop = true;
  }
  L->cv_req.notify_all();
  L->cv_free.notify_all();
  if (L->worker.joinable()) L->worker.join();
  (void)hipStreamSynchronize(L->copy_stream);
  for (Slot& s : L->slots) {
    if (s.release_pending) (void)hipEventSynchronize(s.released);
    if (s.pin) (void)hipHostFree(s.pin);
    if (s.dev) (void)hipFree(s.dev);
    (void)hipEventDestroy(s.copied);
    (void)hipEventDestroy(s.released);
  }
  (void)hipStreamDestroy(L->copy_stream);
  delete L;
}
