// Host side of the host-buffer entry points (rsp_chain_process / _process_detections): what the
// reference-side binding calls (FftMagCfarChainTester.scala:137,145-151 -> JNI -> here).
//
// The batch is cut into chunks that move as a three-stage pipeline on three HIP streams,
//     H2D(k + 1)  ||  kernel(k)  ||  D2H(k - 1),
// so the PCIe link carries input and output at the same time and the kernels hide under it.  Host
// memory the caller got from rsp_host_alloc (hipHostMalloc) or pinned with rsp_host_register is
// DMA'd in place; pageable memory goes through a ring of pinned staging chunks filled / drained
// by a small pool of copy threads (one thread's memcpy is ~10 GB/s, a fifth of the link).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace rsp {

// fixed pool of copy threads: run(n, f) calls f(i) for i < n on the pool AND the caller, returns when all are done
class CopyPool {
 public:
  explicit CopyPool(int workers) {
    for (int i = 0; i < workers; ++i) threads_.emplace_back([this] { loop(); });
  }
  ~CopyPool() {
    {
      std::lock_guard<std::mutex> l(m_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto& t : threads_) t.join();
  }
  int workers() const { return (int)threads_.size(); }

  void run(int n, const std::function<void(int)>& f) {
    if (n <= 0) return;
    {
      std::lock_guard<std::mutex> l(m_);
      job_ = &f;
      next_ = 0;
      count_ = n;
      pending_ = n;
      ++generation_;
    }
    cv_.notify_all();
    work();  // the caller takes pieces too
    std::unique_lock<std::mutex> l(m_);
    done_.wait(l, [this] { return pending_ == 0; });
    job_ = nullptr;
  }

  // dst <- src in pieces of >= 1 MiB, one per thread
  void copy(void* dst, const void* src, size_t bytes) {
    const size_t min_piece = size_t(1) << 20;
    int parts = (int)((bytes + min_piece - 1) / min_piece);
    const int maxp = workers() + 1;
    if (parts > maxp) parts = maxp;
    if (parts <= 1) {
      std::memcpy(dst, src, bytes);
      return;
    }
    const size_t piece = ((bytes / parts) + 63) & ~size_t(63);
    run(parts, [&](int i) {
      const size_t lo = (size_t)i * piece;
      if (lo >= bytes) return;
      const size_t n = lo + piece > bytes || i == parts - 1 ? bytes - lo : piece;
      std::memcpy(static_cast<char*>(dst) + lo, static_cast<const char*>(src) + lo, n);
    });
  }

 private:
  void work() {
    for (;;) {
      const std::function<void(int)>* f;
      int i;
      {
        std::lock_guard<std::mutex> l(m_);
        if (!job_ || next_ >= count_) return;
        f = job_;
        i = next_++;
      }
      (*f)(i);
      {
        std::lock_guard<std::mutex> l(m_);
        if (--pending_ == 0) done_.notify_all();
      }
    }
  }
  void loop() {
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [&] { return stop_ || generation_ != seen; });
        if (stop_) return;
        seen = generation_;
      }
      work();
    }
  }
  std::vector<std::thread> threads_;
  std::mutex m_;
  std::condition_variable cv_, done_;
  const std::function<void(int)>* job_ = nullptr;
  int next_ = 0, count_ = 0, pending_ = 0;
  uint64_t generation_ = 0;
  bool stop_ = false;
};

// is [p, p + bytes) host memory the GPU can DMA in place (hipHostMalloc / hipHostRegister)?
inline bool host_range_pinned(const void* p, size_t bytes) {
  if (!p || !bytes) return false;
  hipPointerAttribute_t a0, a1;
  if (hipPointerGetAttributes(&a0, p) != hipSuccess) {
    (void)hipGetLastError();  // pageable memory: the query fails and leaves a sticky error behind
    return false;
  }
  if (hipPointerGetAttributes(&a1, static_cast<const char*>(p) + bytes - 1) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return a0.type == hipMemoryTypeHost && a1.type == hipMemoryTypeHost;
}

// streams, events and the pinned staging ring of one chain handle; created on first use
struct HostPipe {
  static constexpr int kSlots = 3;  // staging chunks per direction: one being filled, one on the link, one spare
  hipStream_t h2d = nullptr, d2h = nullptr;
  std::vector<hipEvent_t> ev_in, ev_k, ev_out;  // per chunk: input landed / kernel done / output landed (staged path)
  void* stage_in[kSlots] = {};
  void* stage_out[kSlots] = {};
  size_t stage_in_bytes = 0, stage_out_bytes = 0;
  CopyPool* pool = nullptr;

  hipError_t init() {
    if (h2d) return hipSuccess;
    hipError_t e = hipStreamCreateWithFlags(&h2d, hipStreamNonBlocking);
    if (e != hipSuccess) return e;
    return hipStreamCreateWithFlags(&d2h, hipStreamNonBlocking);
  }
  hipError_t events(size_t chunks) {
    while (ev_in.size() < chunks) {
      hipEvent_t a, b, c;
      hipError_t e = hipEventCreateWithFlags(&a, hipEventDisableTiming);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&b, hipEventDisableTiming);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&c, hipEventDisableTiming);
      if (e != hipSuccess) return e;
      ev_in.push_back(a);
      ev_k.push_back(b);
      ev_out.push_back(c);
    }
    return hipSuccess;
  }
  hipError_t staging(bool in, size_t bytes) {
    void** ring = in ? stage_in : stage_out;
    size_t& have = in ? stage_in_bytes : stage_out_bytes;
    if (have >= bytes) return hipSuccess;
    for (int i = 0; i < kSlots; ++i) {
      if (ring[i]) (void)hipHostFree(ring[i]);
      ring[i] = nullptr;
    }
    have = 0;
    for (int i = 0; i < kSlots; ++i) {
      hipError_t e = hipHostMalloc(&ring[i], bytes, hipHostMallocDefault);
      if (e != hipSuccess) return e;
    }
    have = bytes;
    return hipSuccess;
  }
  CopyPool& copier() {
    if (!pool) {
      unsigned hw = std::thread::hardware_concurrency();
      int w = hw >= 16 ? 7 : hw >= 8 ? 5 : hw >= 4 ? 2 : 1;
      pool = new CopyPool(w);
    }
    return *pool;
  }
  void destroy() {
    for (auto v : {&ev_in, &ev_k, &ev_out}) {
      for (hipEvent_t e : *v) (void)hipEventDestroy(e);
      v->clear();
    }
    for (int i = 0; i < kSlots; ++i) {
      if (stage_in[i]) (void)hipHostFree(stage_in[i]);
      if (stage_out[i]) (void)hipHostFree(stage_out[i]);
      stage_in[i] = stage_out[i] = nullptr;
    }
    stage_in_bytes = stage_out_bytes = 0;
    if (h2d) (void)hipStreamDestroy(h2d);
    if (d2h) (void)hipStreamDestroy(d2h);
    h2d = d2h = nullptr;
    delete pool;
    pool = nullptr;
  }
};

}  // namespace rsp
