// libbls381_hip.so -- the C ABI of include/bls381_hip.h (contexts, staging, launches; kernels live in k_*.hip).
//
// One pairing per lane pair (fp2_pair.hip.h); a workgroup is `block` lanes (default one 64-lane wave = 32 pairings).  Each entry point cites
// the reference item it replaces in include/bls381_hip.h.  There is NO host fallback: without a HIP device
// bls381_create fails with BLS381_E_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

#include "../../include/bls381_hip.h"
#include "launch.h"

// ------------------------------------------------------------------------------------------ context
struct bls381_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_mid = nullptr;   // ev_mid: between the Miller loops and the final exponentiation
  bool timed = false, timed_mid = false;
  int io_form = BLS381_IO_ARK_MONT;
  int fp2_shortcut = 0;
  int block = 64;
  int last_hip = 0;
  int fexp_mode = BLS381_FEXP_DEFAULT;   // BLS381_OPT_FEXP_MODE
  uint32_t* raw_scratch = nullptr;   // n x 16 dwords: norms / norm inverses between the final-exponentiation kernels
  size_t raw_scratch_elems = 0;
  uint32_t* stage_scratch = nullptr; // staged form: lane-sliced stage values (csrc/fexp_stages.hip.h)
  size_t stage_scratch_bytes = 0;
  uint64_t* part_scratch = nullptr;  // products of more than three pairs: Miller value of the current group of pairs (n x 72 u64)
  size_t part_scratch_elems = 0;
  hipEvent_t part_done = nullptr;    // recorded behind the last launch that used part_scratch
  hipStream_t part_stream = nullptr;
  bool part_used = false;
  hipEvent_t scratch_done = nullptr; // recorded after the last launch that used raw_scratch
  // staged final exponentiation: the batch runs as two half-batch pipelines on two auxiliary streams, so that the ramp,
  // tail and stalls of one stage kernel (and the latency-bound inversion kernels) overlap with the other half's work
  static constexpr int MAX_PIPES = 4;
  int pipes = 2;                     // BLS381_OPT_FEXP_PIPELINES
  bool pipes_set = false;            // ... set by the caller (else: two pipelines from 2^18 elements, one below)
  int hom_miller = 1;                // BLS381_OPT_PAIRING_HOMOGENEOUS_MILLER
  int fexp_chunk_log2 = 20;          // BLS381_OPT_FEXP_CHUNK_LOG2: staged form, elements per pass (bounds the stage scratch)
  // BLS381_INFO_*: what the most recent call ran (fexp_launch / launch_multi record it)
  int last_fexp_mode = -1, last_fexp_pipes = -1, last_hom = -1, last_group = -1;
  hipStream_t aux[MAX_PIPES] = {};
  hipEvent_t fork = nullptr, stagger[MAX_PIPES] = {}, join[MAX_PIPES] = {};
  hipStream_t scratch_stream = nullptr;
  bool scratch_used = false;
  // host-pointer entry points: context-owned, grow-only device staging (kind: in1, in2, out, flags; two slots for the
  // double-buffered pipeline), copy streams and the events that order copies against launches
  void* stage_buf[4][2] = {};
  size_t stage_cap[4][2] = {};
  hipStream_t h2d = nullptr, d2h = nullptr;
  hipEvent_t in_ready[2] = {}, computed[2] = {}, out_copied[2] = {};
};

#define HIP_TRY(ctx, expr)                                        \
  do {                                                            \
    hipError_t e_ = (expr);                                       \
    if (e_ != hipSuccess) { (ctx)->last_hip = (int)e_; return (e_ == hipErrorOutOfMemory) ? BLS381_E_NOMEM : BLS381_E_HIP; } \
  } while (0)

extern "C" {

const char* bls381_strerror(int code) {
  switch (code) {
    case BLS381_OK: return "ok";
    case BLS381_E_INVALID: return "invalid argument";
    case BLS381_E_NO_DEVICE: return "no usable HIP device (this library has no CPU path)";
    case BLS381_E_HIP: return "HIP runtime error";
    case BLS381_E_NOMEM: return "device allocation failed";
    default: return "unknown error";
  }
}

int bls381_create(bls381_ctx** out, int device_id) {
  if (!out) return BLS381_E_INVALID;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return BLS381_E_NO_DEVICE;
  if (device_id < 0 || device_id >= count) return BLS381_E_INVALID;
  bls381_ctx* c = new (std::nothrow) bls381_ctx();
  if (!c) return BLS381_E_NOMEM;
  c->device = device_id;
  if (hipSetDevice(device_id) != hipSuccess || hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess || hipEventCreate(&c->ev_mid) != hipSuccess ||
      hipEventCreateWithFlags(&c->scratch_done, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->part_done, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->fork, hipEventDisableTiming) != hipSuccess) {
    bls381_destroy(c);
    return BLS381_E_NO_DEVICE;
  }
  for (int i = 0; i < bls381_ctx::MAX_PIPES; ++i) {
    if (hipStreamCreateWithFlags(&c->aux[i], hipStreamNonBlocking) == hipSuccess &&
        hipEventCreateWithFlags(&c->stagger[i], hipEventDisableTiming) == hipSuccess &&
        hipEventCreateWithFlags(&c->join[i], hipEventDisableTiming) == hipSuccess) continue;
    bls381_destroy(c);
    return BLS381_E_NO_DEVICE;
  }
  bool ok = hipStreamCreateWithFlags(&c->h2d, hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&c->d2h, hipStreamNonBlocking) == hipSuccess;
  for (int i = 0; ok && i < 2; ++i)
    ok = hipEventCreateWithFlags(&c->in_ready[i], hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&c->computed[i], hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&c->out_copied[i], hipEventDisableTiming) == hipSuccess;
  if (!ok) { bls381_destroy(c); return BLS381_E_NO_DEVICE; }
  c->stream = c->own_stream;
  *out = c;
  return BLS381_OK;
}

void bls381_destroy(bls381_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->ev_mid) (void)hipEventDestroy(c->ev_mid);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  if (c->raw_scratch) (void)hipFree(c->raw_scratch);
  if (c->stage_scratch) (void)hipFree(c->stage_scratch);
  if (c->part_scratch) (void)hipFree(c->part_scratch);
  if (c->scratch_done) (void)hipEventDestroy(c->scratch_done);
  if (c->part_done) (void)hipEventDestroy(c->part_done);
  for (int i = 0; i < bls381_ctx::MAX_PIPES; ++i) {
    if (c->aux[i]) (void)hipStreamDestroy(c->aux[i]);
    if (c->join[i]) (void)hipEventDestroy(c->join[i]);
    if (c->stagger[i]) (void)hipEventDestroy(c->stagger[i]);
  }
  if (c->fork) (void)hipEventDestroy(c->fork);
  for (int k = 0; k < 4; ++k) for (int i = 0; i < 2; ++i) if (c->stage_buf[k][i]) (void)hipFree(c->stage_buf[k][i]);
  for (int i = 0; i < 2; ++i) {
    if (c->in_ready[i]) (void)hipEventDestroy(c->in_ready[i]);
    if (c->computed[i]) (void)hipEventDestroy(c->computed[i]);
    if (c->out_copied[i]) (void)hipEventDestroy(c->out_copied[i]);
  }
  if (c->h2d) (void)hipStreamDestroy(c->h2d);
  if (c->d2h) (void)hipStreamDestroy(c->d2h);
  delete c;
}

int bls381_set_option(bls381_ctx* c, int option, int value) {
  if (!c) return BLS381_E_INVALID;
  switch (option) {
    case BLS381_OPT_IO_FORM:
      if (value != BLS381_IO_ARK_MONT && value != BLS381_IO_CANONICAL) return BLS381_E_INVALID;
      c->io_form = value; return BLS381_OK;
    case BLS381_OPT_REF_OPT_FP2_SHORTCUT: c->fp2_shortcut = value ? 1 : 0; return BLS381_OK;
    case BLS381_OPT_BLOCK_SIZE:
      if (value != 64 && value != 128 && value != 256) return BLS381_E_INVALID;
      c->block = value; return BLS381_OK;
    case BLS381_OPT_FEXP_MODE:
      if (value < BLS381_FEXP_SELF_CONTAINED || value > BLS381_FEXP_AUTO) return BLS381_E_INVALID;
      c->fexp_mode = value; return BLS381_OK;
    case BLS381_OPT_FEXP_PIPELINES:
      if (value < 0 || value > bls381_ctx::MAX_PIPES) return BLS381_E_INVALID;
      c->pipes = value ? value : 2; c->pipes_set = value != 0; return BLS381_OK;      // 0 = automatic
    case BLS381_OPT_PAIRING_HOMOGENEOUS_MILLER: c->hom_miller = value ? 1 : 0; return BLS381_OK;
    case BLS381_OPT_FEXP_CHUNK_LOG2:
      if (value < 10 || value > 22) return BLS381_E_INVALID;
      c->fexp_chunk_log2 = value; return BLS381_OK;
    default: return BLS381_E_INVALID;
  }
}

int bls381_get_option(const bls381_ctx* c, int option, int* value) {
  if (!c || !value) return BLS381_E_INVALID;
  switch (option) {
    case BLS381_OPT_IO_FORM: *value = c->io_form; return BLS381_OK;
    case BLS381_OPT_REF_OPT_FP2_SHORTCUT: *value = c->fp2_shortcut; return BLS381_OK;
    case BLS381_OPT_BLOCK_SIZE: *value = c->block; return BLS381_OK;
    case BLS381_OPT_FEXP_MODE: *value = c->fexp_mode; return BLS381_OK;
    case BLS381_OPT_FEXP_PIPELINES: *value = c->pipes_set ? c->pipes : 0; return BLS381_OK;
    case BLS381_OPT_PAIRING_HOMOGENEOUS_MILLER: *value = c->hom_miller; return BLS381_OK;
    case BLS381_OPT_FEXP_CHUNK_LOG2: *value = c->fexp_chunk_log2; return BLS381_OK;
    case BLS381_INFO_LAST_FEXP_MODE: *value = c->last_fexp_mode; return BLS381_OK;
    case BLS381_INFO_LAST_FEXP_PIPELINES: *value = c->last_fexp_pipes; return BLS381_OK;
    case BLS381_INFO_LAST_MILLER_HOMOGENEOUS: *value = c->last_hom; return BLS381_OK;
    case BLS381_INFO_LAST_MILLER_GROUP: *value = c->last_group; return BLS381_OK;
    default: return BLS381_E_INVALID;
  }
}

int bls381_set_stream(bls381_ctx* c, void* s) {
  if (!c) return BLS381_E_INVALID;
  c->stream = (hipStream_t)s;          // NULL is a stream too: the legacy default stream
  return BLS381_OK;
}

int bls381_use_own_stream(bls381_ctx* c) {
  if (!c) return BLS381_E_INVALID;
  c->stream = c->own_stream;
  return BLS381_OK;
}

int bls381_sync(bls381_ctx* c) {
  if (!c) return BLS381_E_INVALID;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return BLS381_OK;
}

int bls381_last_hip_error(const bls381_ctx* c) { return c ? c->last_hip : 0; }

float bls381_last_kernel_ms(const bls381_ctx* c) {
  if (!c || !c->timed) return -1.0f;
  if (hipEventSynchronize(c->ev1) != hipSuccess) return -1.0f;
  float ms = -1.0f;
  if (hipEventElapsedTime(&ms, c->ev0, c->ev1) != hipSuccess) return -1.0f;
  return ms;
}

float bls381_last_phase_ms(const bls381_ctx* c, int phase) {
  if (!c || !c->timed || !c->timed_mid || phase < 0 || phase > 1) return -1.0f;
  if (hipEventSynchronize(c->ev1) != hipSuccess) return -1.0f;
  float ms = -1.0f;
  const hipError_t e = phase == 0 ? hipEventElapsedTime(&ms, c->ev0, c->ev_mid) : hipEventElapsedTime(&ms, c->ev_mid, c->ev1);
  return e == hipSuccess ? ms : -1.0f;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------ launches
namespace {

struct Timed {
  bls381_ctx* c;
  explicit Timed(bls381_ctx* c_) : c(c_) { c->timed_mid = false; (void)hipEventRecord(c->ev0, c->stream); }
  ~Timed() { (void)hipEventRecord(c->ev1, c->stream); c->timed = true; }
};

// Device scratch of the multi-kernel final exponentiations: grow-only, owned by the context.  The launches are
// asynchronous, so a previous call may still be using the buffers: on the same stream the order is implicit; after
// bls381_set_stream the new stream first waits for the event recorded behind the last user; an outgrown buffer is released
// only after that user has finished.
constexpr size_t FEXP_AUTO_STAGED_MIN = (size_t)1 << 10; // BLS381_FEXP_AUTO: staged form from this batch size on.  With the binary-GCD
                                                         // inversion it wins at every measured size (tools/time_fexp_modes.py, ms:
                                                         // 2^10 4.8 vs 5.4, 2^13 5.0 vs 5.5, 2^15 5.4 vs 6.4, 2^17 17.8 vs 21.8, 2^20 130 vs 162)
constexpr size_t FEXP_PIPELINE_MIN = (size_t)1 << 14;     // smallest batch an explicit BLS381_OPT_FEXP_PIPELINES setting is applied to
constexpr size_t FEXP_PIPELINE_AUTO_MIN = (size_t)1 << 18; // default: two half-batch pipelines from here on, one below
                                                         // (tools/time_fexp_pipelines.py: 2^15 5.4 vs 7.1 ms, 2^17 17.8 vs 17.9, 2^18 33.9 vs 33.4)
// pipelines for a staged final exponentiation of n elements
int fexp_pipes(const bls381_ctx* c, size_t n) {
  if (c->pipes_set) return n >= FEXP_PIPELINE_MIN ? c->pipes : 1;
  return n >= FEXP_PIPELINE_AUTO_MIN ? 2 : 1;
}

struct FexpPlan { uint32_t* raw = nullptr; uint32_t* stage = nullptr; size_t chunk = 0; int pipes = 1; };
// elements per pipeline and pass: an even share of the chunk, rounded up to a multiple of 128
size_t fexp_part(size_t chunk, int pipes) { return ((chunk + pipes - 1) / pipes + 127) / 128 * 128; }

int fexp_acquire(bls381_ctx* c, size_t n, FexpPlan* plan) {
  *plan = FexpPlan();
  if (c->fexp_mode == BLS381_FEXP_SELF_CONTAINED) return BLS381_OK;
  if (c->scratch_used && c->scratch_stream != c->stream) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->scratch_done, 0));
  bool staged = c->fexp_mode == BLS381_FEXP_STAGED || (c->fexp_mode == BLS381_FEXP_AUTO && n >= FEXP_AUTO_STAGED_MIN);
  // raw scratch always covers the whole batch (the batched-inversion form, also the fallback, is not chunked)
  const size_t max_chunk = (size_t)1 << c->fexp_chunk_log2;   // 2^20 by default: bounds the stage scratch at ~6.1 GB
  const size_t chunk = (staged && n > max_chunk) ? max_chunk : n;
  // staged: every pipeline owns a share of fexp_part(chunk) elements of both scratches
  const int pipes = fexp_pipes(c, n);
  const size_t stage_bytes = staged ? (size_t)pipes * bls_launch::fexp_stage_scratch_bytes(fexp_part(chunk, pipes)) : 0;
  const size_t raw_elems = staged ? (n > (size_t)pipes * fexp_part(chunk, pipes) ? n : (size_t)pipes * fexp_part(chunk, pipes)) : n;
  if (c->raw_scratch_elems < raw_elems || c->stage_scratch_bytes < stage_bytes) {
    if (c->scratch_used) HIP_TRY(c, hipEventSynchronize(c->scratch_done));
    if (c->raw_scratch_elems < raw_elems) {
      if (c->raw_scratch) { (void)hipFree(c->raw_scratch); c->raw_scratch = nullptr; c->raw_scratch_elems = 0; }
      HIP_TRY(c, hipMalloc((void**)&c->raw_scratch, raw_elems * 16 * sizeof(uint32_t)));
      c->raw_scratch_elems = raw_elems;
    }
    if (c->stage_scratch_bytes < stage_bytes) {
      if (c->stage_scratch) { (void)hipFree(c->stage_scratch); c->stage_scratch = nullptr; c->stage_scratch_bytes = 0; }
      const hipError_t e = hipMalloc((void**)&c->stage_scratch, stage_bytes);
      if (e == hipErrorOutOfMemory) {            // no room for the stage scratch: this call runs the batched-inversion form
        (void)hipGetLastError();
        c->stage_scratch = nullptr;
        staged = false;
      } else {
        HIP_TRY(c, e);
        c->stage_scratch_bytes = stage_bytes;
      }
    }
  }
  plan->raw = c->raw_scratch;
  plan->stage = staged ? c->stage_scratch : nullptr;
  plan->chunk = chunk;
  plan->pipes = pipes;
  return BLS381_OK;
}
// the final exponentiation of `n` elements in the context's mode, queued on c->stream (in place allowed)
int fexp_launch(bls381_ctx* c, const FexpPlan& plan, const uint64_t* in, uint64_t* out, uint8_t* flags, size_t n) {
  c->last_fexp_mode = plan.stage ? BLS381_FEXP_STAGED : (plan.raw ? BLS381_FEXP_BATCHED_INVERSION : BLS381_FEXP_SELF_CONTAINED);
  c->last_fexp_pipes = plan.stage ? plan.pipes : 1;
  if (plan.stage && plan.pipes > 1) {
    // `pipes` part-batch pipelines: fork from the caller's stream, each part on its own stream and its own share of the
    // scratch, join back.  Parts are cut at multiples of 128 elements so that every launch shape fits its scratch share.
    // Pipeline p starts when pipeline p-1 has its stage 1 through: from then on they are a beat apart, and each one's
    // inversion kernels, ramps and tails fall into the others' stage kernels.
    const int P = plan.pipes;
    const size_t part_chunk = fexp_part(plan.chunk, P);           // elements per pipeline and pass (scratch share)
    const size_t part = fexp_part(n, P);                          // elements per pipeline overall
    const size_t stage_dwords_part = bls_launch::fexp_stage_scratch_bytes(part_chunk) / sizeof(uint32_t);
    HIP_TRY(c, hipEventRecord(c->fork, c->stream));
    for (int h = 0; h < P; ++h) {
      const size_t first = (size_t)h * part < n ? (size_t)h * part : n;
      const size_t cnt = (h == P - 1) ? n - first : (first + part < n ? part : n - first);
      hipEvent_t record_after[6] = {c->stagger[h]};                // behind this pipeline's stage 1
      HIP_TRY(c, hipStreamWaitEvent(c->aux[h], h ? c->stagger[h - 1] : c->fork, 0));
      if (cnt) HIP_TRY(c, bls_launch::final_exp_staged(in + 72 * first, out + 72 * first, flags ? flags + first : nullptr, cnt, c->io_form, c->block,
                                                       c->aux[h], plan.raw + (size_t)h * part_chunk * 16, plan.stage + (size_t)h * stage_dwords_part, part_chunk,
                                                       nullptr, record_after));
      else HIP_TRY(c, hipEventRecord(c->stagger[h], c->aux[h]));
      HIP_TRY(c, hipEventRecord(c->join[h], c->aux[h]));
      HIP_TRY(c, hipStreamWaitEvent(c->stream, c->join[h], 0));
    }
  } else if (plan.stage) {
    HIP_TRY(c, bls_launch::final_exp_staged(in, out, flags, n, c->io_form, c->block, c->stream, plan.raw, plan.stage, plan.chunk, nullptr, nullptr));
  } else {
    bls_launch::final_exp(in, out, flags, n, c->io_form, c->block, c->stream, plan.raw);
  }
  if (plan.raw) {                                     // mark the scratch busy until this point of the stream
    HIP_TRY(c, hipEventRecord(c->scratch_done, c->stream));
    c->scratch_stream = c->stream;
    c->scratch_used = true;
  }
  return BLS381_OK;
}

int launch_ref_opt(bls381_ctx* c, const uint64_t* g1, const uint64_t* g2, uint64_t* out, uint8_t* flags, size_t n) {
  if (n == 0) return BLS381_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  {
    Timed t(c);
    c->last_hom = -1; c->last_group = -1; c->last_fexp_mode = -1; c->last_fexp_pipes = -1;
    bls_launch::ref_opt(c->fp2_shortcut != 0, g1, g2, out, flags, n, c->io_form, c->block, c->stream);
  }
  HIP_TRY(c, hipGetLastError());
  return BLS381_OK;
}

// Pairs per Miller launch for a product of k pairs: up to three pairs go in one launch (shared squarings); longer products
// are cut into groups of three (round 3 measured four pairs in one launch -- a kernel that calls its step routines -- at 34.0 ms per
// 2^16 against 31.7 ms for 3 + 1; tools/time_multi_k.py), the groups' Miller values are multiplied
// on the device (bit-identical to one shared loop: squaring distributes over the product) and ONE final exponentiation
// follows -- `multi_miller_loop(terms: &[..])` takes a slice of any length (src/miller_loop_native.rs:154-212).
constexpr size_t MULTI_GROUP = 3;

int part_acquire(bls381_ctx* c, size_t n) {
  if (c->part_used && c->part_stream != c->stream) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->part_done, 0));
  if (c->part_scratch_elems >= n) return BLS381_OK;
  if (c->part_used) HIP_TRY(c, hipEventSynchronize(c->part_done));
  if (c->part_scratch) { (void)hipFree(c->part_scratch); c->part_scratch = nullptr; c->part_scratch_elems = 0; }
  HIP_TRY(c, hipMalloc((void**)&c->part_scratch, n * 72 * sizeof(uint64_t)));
  c->part_scratch_elems = n;
  return BLS381_OK;
}

template <bool FE>
int launch_multi(bls381_ctx* c, const uint64_t* g1, const uint64_t* g2, size_t k, uint64_t* out, uint8_t* flags, size_t n) {
  if (n == 0) return BLS381_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  FexpPlan plan;
  if (FE) { const int rc = fexp_acquire(c, n, &plan); if (rc != BLS381_OK) return rc; }
  if (k > MULTI_GROUP) { const int rc = part_acquire(c, n); if (rc != BLS381_OK) return rc; }
  int rc = BLS381_OK;
  {
    Timed t(c);
    const bool hom = FE && c->hom_miller;
    c->last_hom = hom ? 1 : 0;
    c->last_group = (int)(k <= MULTI_GROUP ? k : MULTI_GROUP);
    if (!FE) { c->last_fexp_mode = -1; c->last_fexp_pipes = -1; }
    if (k <= MULTI_GROUP) {
      if (!bls_launch::multi(k, hom, g1, g2, k, out, flags, false, n, c->io_form, c->block, c->stream)) return BLS381_E_INVALID;
    } else {
      for (size_t first = 0; first < k; first += MULTI_GROUP) {
        const size_t kc = (k - first < MULTI_GROUP) ? k - first : MULTI_GROUP;
        uint64_t* dst = first ? c->part_scratch : out;
        if (!bls_launch::multi(kc, hom, g1 + 12 * first, g2 + 24 * first, k, dst, flags, first != 0, n, c->io_form, c->block, c->stream))
          return BLS381_E_INVALID;
        if (first) bls_launch::fp12_op(BLS381_FP12_MUL, out, c->part_scratch, out, n, c->io_form, c->block, c->stream);
      }
      HIP_TRY(c, hipEventRecord(c->part_done, c->stream));
      c->part_stream = c->stream;
      c->part_used = true;
    }
    if (FE) {
      c->timed_mid = hipEventRecord(c->ev_mid, c->stream) == hipSuccess;      // bls381_last_phase_ms: Miller loops | final exponentiation
      rc = fexp_launch(c, plan, out, out, nullptr, n);
    }
  }
  if (rc != BLS381_OK) return rc;
  HIP_TRY(c, hipGetLastError());
  return BLS381_OK;
}

int launch_final_exp(bls381_ctx* c, const uint64_t* in, uint64_t* out, uint8_t* flags, size_t n) {
  if (n == 0) return BLS381_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  FexpPlan plan;
  { const int rc = fexp_acquire(c, n, &plan); if (rc != BLS381_OK) return rc; }
  int rc;
  {
    Timed t(c);
    rc = fexp_launch(c, plan, in, out, flags, n);
  }
  if (rc != BLS381_OK) return rc;
  HIP_TRY(c, hipGetLastError());
  return BLS381_OK;
}

// RAII device buffers for the host-pointer entry points
struct DevBuf {
  void* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
};

// Device staging of the host-pointer entry points: grow-only buffers owned by the context (no hipMalloc per call once warm).
// Every host-pointer entry point ends with a synchronisation, so the buffers are idle whenever one starts.
enum { ST_IN1 = 0, ST_IN2 = 1, ST_OUT = 2, ST_FLAGS = 3 };
int stage_ensure(bls381_ctx* c, int kind, int slot, size_t bytes, void** p) {
  *p = nullptr;
  if (bytes == 0) return BLS381_OK;
  if (c->stage_cap[kind][slot] < bytes) {
    if (c->stage_buf[kind][slot]) { (void)hipFree(c->stage_buf[kind][slot]); c->stage_buf[kind][slot] = nullptr; c->stage_cap[kind][slot] = 0; }
    HIP_TRY(c, hipMalloc(&c->stage_buf[kind][slot], bytes));
    c->stage_cap[kind][slot] = bytes;
  }
  *p = c->stage_buf[kind][slot];
  return BLS381_OK;
}

// Stages host inputs to the device, runs `body` on device pointers, copies the outputs back (one piece, for the small surfaces).
template <class Body>
int with_staging_queue(bls381_ctx* c, const void* in1, size_t in1_bytes, const void* in2, size_t in2_bytes,
                       void* out, size_t out_bytes, uint8_t* flags, size_t n, Body& body) {
  void *d1, *d2, *dout, *dfl;
  int rc;
  if ((rc = stage_ensure(c, ST_IN1, 0, in1_bytes, &d1)) || (rc = stage_ensure(c, ST_IN2, 0, in2 ? in2_bytes : 0, &d2)) ||
      (rc = stage_ensure(c, ST_OUT, 0, out_bytes, &dout)) || (rc = stage_ensure(c, ST_FLAGS, 0, flags ? n : 0, &dfl))) return rc;
  if (d1) HIP_TRY(c, hipMemcpyAsync(d1, in1, in1_bytes, hipMemcpyHostToDevice, c->stream));
  if (d2) HIP_TRY(c, hipMemcpyAsync(d2, in2, in2_bytes, hipMemcpyHostToDevice, c->stream));
  rc = body((const uint64_t*)d1, (const uint64_t*)d2, (uint64_t*)dout, (uint8_t*)dfl);
  if (rc != BLS381_OK) return rc;
  HIP_TRY(c, hipMemcpyAsync(out, dout, out_bytes, hipMemcpyDeviceToHost, c->stream));
  if (flags) HIP_TRY(c, hipMemcpyAsync(flags, dfl, n, hipMemcpyDeviceToHost, c->stream));
  return BLS381_OK;
}
// Whatever happens while the copies and launches are queued, the call returns only after the stream has drained: queued copies read
// and write the CALLER's host buffers, which the caller may free or reuse as soon as it sees the return value, and the next call
// assumes the staging buffers are idle.
template <class Body>
int with_staging(bls381_ctx* c, const void* in1, size_t in1_bytes, const void* in2, size_t in2_bytes,
                 void* out, size_t out_bytes, uint8_t* flags, size_t n, Body body) {
  HIP_TRY(c, hipSetDevice(c->device));
  const int rc = with_staging_queue(c, in1, in1_bytes, in2, in2_bytes, out, out_bytes, flags, n, body);
  const hipError_t e = hipStreamSynchronize(c->stream);
  if (rc != BLS381_OK) return rc;
  HIP_TRY(c, e);
  return BLS381_OK;
}

// The pairing-sized entry points: the batch moves in pieces of STAGE_PIECE elements through two device slots, copies on their own
// streams, so that the upload of piece i+1 and the download of piece i-1 run under the kernels of piece i.  What stays exposed is
// the first upload and the last download.  A piece (2^18 elements) is large enough for the staged final exponentiation to run
// within 3 % of its full-batch rate and small enough that the exposed copies are 75 MB up and 151 MB down.
// w1 / w2 / wo: bytes per element of the two inputs and of the output; body(in1, in2, out, flags, first, m) queues m elements on c->stream.
// 2^18: bench.py --from-host at 2^20 pairings 262.5 ms against 265.1 (2^19) and 270.8 (2^17); 254 ms resident
static int stage_piece_log2() {
  const char* e = getenv("BLS381_STAGE_PIECE_LOG2");                 // A/B switch, clamped to 2^14 .. 2^22
  const int v = e ? atoi(e) : 18;
  return v < 14 ? 14 : (v > 22 ? 22 : v);
}
static const size_t STAGE_PIECE = (size_t)1 << stage_piece_log2();
template <class Body>
int staged_pipeline_queue(bls381_ctx* c, const void* in1, size_t w1, const void* in2, size_t w2, void* out, size_t wo, uint8_t* flags, size_t n, Body& body) {
  const size_t piece = (n <= STAGE_PIECE + STAGE_PIECE / 2) ? n : STAGE_PIECE;
  const size_t pieces = (n + piece - 1) / piece;
  const int slots = pieces > 1 ? 2 : 1;
  void *d1[2] = {}, *d2[2] = {}, *dout[2] = {}, *dfl[2] = {};
  for (int sl = 0; sl < slots; ++sl) {
    int rc;
    if ((rc = stage_ensure(c, ST_IN1, sl, piece * w1, &d1[sl])) || (rc = stage_ensure(c, ST_IN2, sl, in2 ? piece * w2 : 0, &d2[sl])) ||
        (rc = stage_ensure(c, ST_OUT, sl, piece * wo, &dout[sl])) || (rc = stage_ensure(c, ST_FLAGS, sl, flags ? piece : 0, &dfl[sl]))) return rc;
  }
  auto download = [&](size_t i) -> int {
    const int sl = (int)(i & 1);
    const size_t first = i * piece, m = (n - first < piece) ? n - first : piece;
    HIP_TRY(c, hipStreamWaitEvent(c->d2h, c->computed[sl], 0));
    HIP_TRY(c, hipMemcpyAsync((char*)out + first * wo, dout[sl], m * wo, hipMemcpyDeviceToHost, c->d2h));
    if (flags) HIP_TRY(c, hipMemcpyAsync(flags + first, dfl[sl], m, hipMemcpyDeviceToHost, c->d2h));
    HIP_TRY(c, hipEventRecord(c->out_copied[sl], c->d2h));
    return BLS381_OK;
  };
  for (size_t i = 0; i < pieces; ++i) {
    const int sl = (int)(i & 1);
    const size_t first = i * piece, m = (n - first < piece) ? n - first : piece;
    if (i >= 2) HIP_TRY(c, hipStreamWaitEvent(c->h2d, c->computed[sl], 0));        // the kernels of piece i-2 have read this slot
    if (d1[sl]) HIP_TRY(c, hipMemcpyAsync(d1[sl], (const char*)in1 + first * w1, m * w1, hipMemcpyHostToDevice, c->h2d));
    if (d2[sl]) HIP_TRY(c, hipMemcpyAsync(d2[sl], (const char*)in2 + first * w2, m * w2, hipMemcpyHostToDevice, c->h2d));
    HIP_TRY(c, hipEventRecord(c->in_ready[sl], c->h2d));
    HIP_TRY(c, hipStreamWaitEvent(c->stream, c->in_ready[sl], 0));
    if (i >= 2) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->out_copied[sl], 0));    // piece i-2 has left the output slot
    const int rc = body((const uint64_t*)d1[sl], (const uint64_t*)d2[sl], (uint64_t*)dout[sl], (uint8_t*)dfl[sl], first, m);
    if (rc != BLS381_OK) return rc;
    HIP_TRY(c, hipEventRecord(c->computed[sl], c->stream));
    if (i >= 1) { const int r2 = download(i - 1); if (r2 != BLS381_OK) return r2; }   // blocks the host at most until piece i-1 is computed; piece i is queued
  }
  return download(pieces - 1);
}
// One exit for every outcome (see with_staging): the three streams are drained before the call returns, also after an error in the
// middle of the loop -- copies already queued on h2d / d2h / stream touch the caller's buffers and the context's two slots.
template <class Body>
int staged_pipeline(bls381_ctx* c, const void* in1, size_t w1, const void* in2, size_t w2, void* out, size_t wo, uint8_t* flags, size_t n, Body body) {
  HIP_TRY(c, hipSetDevice(c->device));
  const int rc = staged_pipeline_queue(c, in1, w1, in2, w2, out, wo, flags, n, body);
  const hipError_t e0 = hipStreamSynchronize(c->h2d), e1 = hipStreamSynchronize(c->stream), e2 = hipStreamSynchronize(c->d2h);
  if (rc != BLS381_OK) return rc;
  HIP_TRY(c, e0); HIP_TRY(c, e1); HIP_TRY(c, e2);
  return BLS381_OK;
}

}  // namespace

extern "C" {

int bls381_miller_ref_opt_batch_dev(bls381_ctx* c, const uint64_t* g1, const uint64_t* g2, uint64_t* out, uint8_t* flags, size_t n) {
  if (!c || (n && (!g1 || !g2 || !out))) return BLS381_E_INVALID;
  return launch_ref_opt(c, g1, g2, out, flags, n);
}

int bls381_miller_ref_opt_batch(bls381_ctx* c, const uint64_t* g1, const uint64_t* g2, uint64_t* out, uint8_t* flags, size_t n) {
  if (!c || (n && (!g1 || !g2 || !out))) return BLS381_E_INVALID;
  if (n == 0) return BLS381_OK;
  return staged_pipeline(c, g1, 144, g2, 288, out, 576, flags, n,
                         [&](const uint64_t* a, const uint64_t* b, uint64_t* o, uint8_t* f, size_t, size_t m) { return launch_ref_opt(c, a, b, o, f, m); });
}

int bls381_multi_miller_batch_dev(bls381_ctx* c, const uint64_t* g1, const uint64_t* g2, size_t k, uint64_t* out, uint8_t* flags, size_t n) {
  if (!c || k < 1 || (n && (!g1 || !g2 || !out))) return BLS381_E_INVALID;
  return launch_multi<false>(c, g1, g2, k, out, flags, n);
}

int bls381_multi_pairing_batch_dev(bls381_ctx* c, const uint64_t* g1, const uint64_t* g2, size_t k, uint64_t* out, uint8_t* flags, size_t n) {
  if (!c || k < 1 || (n && (!g1 || !g2 || !out))) return BLS381_E_INVALID;
  return launch_multi<true>(c, g1, g2, k, out, flags, n);
}

int bls381_multi_miller_batch(bls381_ctx* c, const uint64_t* g1, const uint64_t* g2, size_t k, uint64_t* out, uint8_t* flags, size_t n) {
  if (!c || k < 1 || (n && (!g1 || !g2 || !out))) return BLS381_E_INVALID;
  if (n == 0) return BLS381_OK;
  return staged_pipeline(c, g1, k * 96, g2, k * 192, out, 576, flags, n,
                         [&](const uint64_t* a, const uint64_t* b, uint64_t* o, uint8_t* f, size_t, size_t m) { return launch_multi<false>(c, a, b, k, o, f, m); });
}

int bls381_multi_pairing_batch(bls381_ctx* c, const uint64_t* g1, const uint64_t* g2, size_t k, uint64_t* out, uint8_t* flags, size_t n) {
  if (!c || k < 1 || (n && (!g1 || !g2 || !out))) return BLS381_E_INVALID;
  if (n == 0) return BLS381_OK;
  return staged_pipeline(c, g1, k * 96, g2, k * 192, out, 576, flags, n,
                         [&](const uint64_t* a, const uint64_t* b, uint64_t* o, uint8_t* f, size_t, size_t m) { return launch_multi<true>(c, a, b, k, o, f, m); });
}

int bls381_miller_ate_batch(bls381_ctx* c, const uint64_t* g1, const uint64_t* g2, uint64_t* out, uint8_t* flags, size_t n) {
  return bls381_multi_miller_batch(c, g1, g2, 1, out, flags, n);
}

int bls381_pairing_batch(bls381_ctx* c, const uint64_t* g1, const uint64_t* g2, uint64_t* out, uint8_t* flags, size_t n) {
  return bls381_multi_pairing_batch(c, g1, g2, 1, out, flags, n);
}

int bls381_final_exp_batch_dev(bls381_ctx* c, const uint64_t* in, uint64_t* out, uint8_t* flags, size_t n) {
  if (!c || (n && (!in || !out))) return BLS381_E_INVALID;
  return launch_final_exp(c, in, out, flags, n);
}

int bls381_final_exp_batch(bls381_ctx* c, const uint64_t* in, uint64_t* out, uint8_t* flags, size_t n) {
  if (!c || (n && (!in || !out))) return BLS381_E_INVALID;
  if (n == 0) return BLS381_OK;
  return staged_pipeline(c, in, 576, nullptr, 0, out, 576, flags, n,
                         [&](const uint64_t* a, const uint64_t*, uint64_t* o, uint8_t* f, size_t, size_t m) { return launch_final_exp(c, a, o, f, m); });
}

int bls381_fp_op_batch(bls381_ctx* c, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
  if (!c || op < 0 || op > BLS381_FP_SGN0 || (n && (!a || !out))) return BLS381_E_INVALID;
  if ((op == BLS381_FP_ADD || op == BLS381_FP_SUB || op == BLS381_FP_MUL || op == BLS381_FP_DIV) && n && !b) return BLS381_E_INVALID;
  if (n == 0) return BLS381_OK;
  return with_staging(c, a, n * 48, b, b ? n * 48 : 0, out, n * 48, nullptr, n,
                      [&](const uint64_t* da, const uint64_t* db, uint64_t* o, uint8_t*) -> int {
                        bls_launch::fp_op(op, da, db, o, n, c->io_form, c->block, c->stream);
                        HIP_TRY(c, hipGetLastError());
                        return (int)BLS381_OK;
                      });
}

int bls381_fp2_op_batch(bls381_ctx* c, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
  if (!c || op < 0 || op > BLS381_FP2_MUL_BY_NONRESIDUE || (n && (!a || !out))) return BLS381_E_INVALID;
  if ((op == BLS381_FP2_ADD || op == BLS381_FP2_SUB || op == BLS381_FP2_MUL) && n && !b) return BLS381_E_INVALID;
  if (n == 0) return BLS381_OK;
  return with_staging(c, a, n * 96, b, b ? n * 96 : 0, out, n * 96, nullptr, n,
                      [&](const uint64_t* da, const uint64_t* db, uint64_t* o, uint8_t*) -> int {
                        bls_launch::fp2_op(op, da, db, o, n, c->io_form, c->block, c->stream);
                        HIP_TRY(c, hipGetLastError());
                        return (int)BLS381_OK;
                      });
}

int bls381_fp6_op_batch(bls381_ctx* c, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
  if (!c || op < 0 || op > BLS381_FP6_MUL_BY_1 || (n && (!a || !out))) return BLS381_E_INVALID;
  if ((op == BLS381_FP6_ADD || op == BLS381_FP6_SUB || op == BLS381_FP6_MUL || op == BLS381_FP6_MUL_BY_01 || op == BLS381_FP6_MUL_BY_1) && n && !b)
    return BLS381_E_INVALID;
  if (n == 0) return BLS381_OK;
  return with_staging(c, a, n * 288, b, b ? n * 288 : 0, out, n * 288, nullptr, n,
                      [&](const uint64_t* da, const uint64_t* db, uint64_t* o, uint8_t*) -> int {
                        bls_launch::fp6_op(op, da, db, o, n, c->io_form, c->block, c->stream);
                        HIP_TRY(c, hipGetLastError());
                        return (int)BLS381_OK;
                      });
}

int bls381_fp12_op_batch(bls381_ctx* c, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
  if (!c || op < 0 || op > BLS381_FP12_MUL_BY_014_KARATSUBA || (n && (!a || !out))) return BLS381_E_INVALID;
  if (op == BLS381_FP12_MUL && n && !b) return BLS381_E_INVALID;
  if (n == 0) return BLS381_OK;
  return with_staging(c, a, n * 576, b, b ? n * 576 : 0, out, n * 576, nullptr, n,
                      [&](const uint64_t* da, const uint64_t* db, uint64_t* o, uint8_t*) -> int {
                        bls_launch::fp12_op(op, da, db, o, n, c->io_form, c->block, c->stream);
                        HIP_TRY(c, hipGetLastError());
                        return (int)BLS381_OK;
                      });
}

int bls381_debug_line_ref_opt_batch(bls381_ctx* c, const uint64_t* q1, const uint64_t* q2, const uint64_t* p, uint64_t* out, size_t n) {
  if (!c || (n && (!q1 || !q2 || !p || !out))) return BLS381_E_INVALID;
  if (n == 0) return BLS381_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  DevBuf dp;
  HIP_TRY(c, hipMalloc(&dp.p, n * 144));
  HIP_TRY(c, hipMemcpyAsync(dp.p, p, n * 144, hipMemcpyHostToDevice, c->stream));
  return with_staging(c, q1, n * 288, q2, n * 288, out, n * 192, nullptr, n,
                      [&](const uint64_t* da, const uint64_t* db, uint64_t* o, uint8_t*) -> int {
                        bls_launch::line_ref_opt(da, db, (const uint64_t*)dp.p, o, n, c->io_form, c->block, c->stream);
                        HIP_TRY(c, hipGetLastError());
                        return (int)BLS381_OK;
                      });
}

int bls381_prepared_pairing_batch_dev(bls381_ctx* c, const uint64_t* g1_aff, const uint64_t* g2_var, const uint64_t* prepared_fixed,
                                      size_t n_var, size_t n_fixed, uint64_t* out_gt, uint8_t* flags, size_t n) {
  if (!c || n_var > 1 || n_fixed < 1 || n_fixed > 2 || (n && (!g1_aff || (n_var && !g2_var) || !prepared_fixed || !out_gt))) return BLS381_E_INVALID;
  if (n == 0) return BLS381_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  FexpPlan plan;
  { const int rc = fexp_acquire(c, n, &plan); if (rc != BLS381_OK) return rc; }
  int rc;
  {
    Timed t(c);
    c->last_hom = c->hom_miller ? 1 : 0; c->last_group = (int)(n_var + n_fixed);
    if (!bls_launch::prepared((int)n_var, (int)n_fixed, c->hom_miller != 0, g1_aff, g2_var, prepared_fixed, out_gt, flags, n, c->io_form, c->block, c->stream))
      return BLS381_E_INVALID;
    c->timed_mid = hipEventRecord(c->ev_mid, c->stream) == hipSuccess;
    rc = fexp_launch(c, plan, out_gt, out_gt, nullptr, n);
  }
  if (rc != BLS381_OK) return rc;
  HIP_TRY(c, hipGetLastError());
  return BLS381_OK;
}

int bls381_prepared_pairing_batch(bls381_ctx* c, const uint64_t* g1_aff, const uint64_t* g2_var, const uint64_t* prepared_fixed,
                                  size_t n_var, size_t n_fixed, uint64_t* out_gt, uint8_t* flags, size_t n) {
  if (!c || n_var > 1 || n_fixed < 1 || n_fixed > 2 || (n && (!g1_aff || (n_var && !g2_var) || !prepared_fixed || !out_gt))) return BLS381_E_INVALID;
  if (n == 0) return BLS381_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  DevBuf dprep;
  const size_t prep_bytes = n_fixed * 68 * 288;
  HIP_TRY(c, hipMalloc(&dprep.p, prep_bytes));
  HIP_TRY(c, hipMemcpyAsync(dprep.p, prepared_fixed, prep_bytes, hipMemcpyHostToDevice, c->stream));
  return staged_pipeline(c, g1_aff, (n_var + n_fixed) * 96, n_var ? g2_var : nullptr, n_var ? 192 : 0, out_gt, 576, flags, n,
                         [&](const uint64_t* a, const uint64_t* b, uint64_t* o, uint8_t* f, size_t, size_t m) -> int {
                           return bls381_prepared_pairing_batch_dev(c, a, b, (const uint64_t*)dprep.p, n_var, n_fixed, o, f, m);
                         });
}

int bls381_groth16_pairing_batch_dev(bls381_ctx* c, const uint64_t* g1_aff, const uint64_t* g2_var, const uint64_t* prepared_fixed,
                                     uint64_t* out_gt, uint8_t* flags, size_t n) {
  return bls381_prepared_pairing_batch_dev(c, g1_aff, g2_var, prepared_fixed, 1, 2, out_gt, flags, n);
}

int bls381_groth16_pairing_batch(bls381_ctx* c, const uint64_t* g1_aff, const uint64_t* g2_var, const uint64_t* prepared_fixed,
                                 uint64_t* out_gt, uint8_t* flags, size_t n) {
  return bls381_prepared_pairing_batch(c, g1_aff, g2_var, prepared_fixed, 1, 2, out_gt, flags, n);
}

int bls381_fp_pow_batch(bls381_ctx* c, const uint64_t* a, const uint64_t* exp, size_t exp_words, uint64_t* out, uint8_t* flags, size_t n) {
  if (!c || !exp || exp_words < 1 || exp_words > 8 || (n && (!a || !out))) return BLS381_E_INVALID;
  if (n == 0) return BLS381_OK;
  return with_staging(c, a, n * 48, nullptr, 0, out, n * 48, flags, n,
                      [&](const uint64_t* da, const uint64_t*, uint64_t* o, uint8_t* f) -> int {
                        bls_launch::fp_pow(da, exp, (int)exp_words, o, f, n, c->io_form, c->block, c->stream);
                        HIP_TRY(c, hipGetLastError());
                        return (int)BLS381_OK;
                      });
}

int bls381_g2_prepare_batch(bls381_ctx* c, const uint64_t* g2_aff, uint64_t* coeffs, uint8_t* flags, size_t n) {
  if (!c || (n && (!g2_aff || !coeffs))) return BLS381_E_INVALID;
  if (n == 0) return BLS381_OK;
  return with_staging(c, g2_aff, n * 192, nullptr, 0, coeffs, n * 68 * 288, flags, n,
                      [&](const uint64_t* dq, const uint64_t*, uint64_t* o, uint8_t* f) -> int {
                        bls_launch::g2_prepare(dq, o, f, n, c->io_form, c->block, c->stream);
                        HIP_TRY(c, hipGetLastError());
                        return (int)BLS381_OK;
                      });
}

// ---- point surface (G1 / G2 Jacobian points)
static int point_op(bls381_ctx* c, int group, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, uint8_t* flags, size_t n) {
  const bool binary = op == BLS381_POINT_ADD || op == BLS381_POINT_EQ;
  const bool predicate = op == BLS381_POINT_EQ || op == BLS381_POINT_IS_ON_CURVE || op == BLS381_POINT_IS_IN_SUBGROUP;
  if (!c || op < BLS381_POINT_ADD || op > BLS381_POINT_IS_IN_SUBGROUP) return BLS381_E_INVALID;
  if (n && (!a || (binary && !b) || (predicate ? !flags : !out))) return BLS381_E_INVALID;
  if (n == 0) return BLS381_OK;
  const size_t bytes = n * (group == 1 ? 144 : 288);
  if (predicate) out = nullptr;                              // predicates answer in flags only
  if (!binary) b = nullptr;
  HIP_TRY(c, hipSetDevice(c->device));
  DevBuf da, db, dout, dfl;
  HIP_TRY(c, hipMalloc(&da.p, bytes));
  HIP_TRY(c, hipMemcpyAsync(da.p, a, bytes, hipMemcpyHostToDevice, c->stream));
  if (b) { HIP_TRY(c, hipMalloc(&db.p, bytes)); HIP_TRY(c, hipMemcpyAsync(db.p, b, bytes, hipMemcpyHostToDevice, c->stream)); }
  if (out) HIP_TRY(c, hipMalloc(&dout.p, bytes));
  if (flags) HIP_TRY(c, hipMalloc(&dfl.p, n));
  bls_launch::point_op(group, op, (const uint64_t*)da.p, (const uint64_t*)db.p, (uint64_t*)dout.p, (uint8_t*)dfl.p, n, c->io_form, c->block, c->stream);
  HIP_TRY(c, hipGetLastError());
  if (out) HIP_TRY(c, hipMemcpyAsync(out, dout.p, bytes, hipMemcpyDeviceToHost, c->stream));
  if (flags) HIP_TRY(c, hipMemcpyAsync(flags, dfl.p, n, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return BLS381_OK;
}
int bls381_g1_op_batch(bls381_ctx* c, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, uint8_t* flags, size_t n) {
  return point_op(c, 1, op, a, b, out, flags, n);
}
int bls381_g2_op_batch(bls381_ctx* c, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, uint8_t* flags, size_t n) {
  return point_op(c, 2, op, a, b, out, flags, n);
}

// ---- point decompression
static int decompress(bls381_ctx* c, int group, const uint64_t* x, const uint8_t* sign, uint64_t* out_aff, uint8_t* flags, size_t n) {
  if (!c || (n && (!x || !sign || !out_aff))) return BLS381_E_INVALID;
  if (n == 0) return BLS381_OK;
  const size_t w = group == 1 ? 48 : 96;
  return with_staging(c, x, n * w, sign, n, out_aff, n * 2 * w, flags, n,
                      [&](const uint64_t* dx, const uint64_t* dsgn, uint64_t* o, uint8_t* f) -> int {
                        bls_launch::decompress(group, dx, (const uint8_t*)dsgn, o, f, n, c->io_form, c->block, c->stream);
                        HIP_TRY(c, hipGetLastError());
                        return (int)BLS381_OK;
                      });
}
int bls381_g1_decompress_batch(bls381_ctx* c, const uint64_t* x, const uint8_t* sign, uint64_t* out_aff, uint8_t* flags, size_t n) {
  return decompress(c, 1, x, sign, out_aff, flags, n);
}
int bls381_g2_decompress_batch(bls381_ctx* c, const uint64_t* x, const uint8_t* sign, uint64_t* out_aff, uint8_t* flags, size_t n) {
  return decompress(c, 2, x, sign, out_aff, flags, n);
}

// ---- witness-generator backend (canonical 12 x u32 digits == 6 x u64 per Fp)
static int witness_inverse(bls381_ctx* c, int op, size_t fp_per_elem, const uint32_t* x, uint32_t* inv, uint8_t* flags, size_t n) {
  if (!c || (n && (!x || !inv))) return BLS381_E_INVALID;
  if (n == 0) return BLS381_OK;
  const size_t bytes = n * fp_per_elem * 48;
  return with_staging(c, x, bytes, nullptr, 0, inv, bytes, flags, n,
                      [&](const uint64_t* dx, const uint64_t*, uint64_t* o, uint8_t* f) -> int {
                        bls_launch::witness_inverse(op, dx, o, f, n, c->block, c->stream);
                        HIP_TRY(c, hipGetLastError());
                        return (int)BLS381_OK;
                      });
}
int bls381_witness_fq12_inverse_batch(bls381_ctx* c, const uint32_t* x, uint32_t* inv, uint8_t* flags, size_t n) {
  return witness_inverse(c, 0, 12, x, inv, flags, n);
}
int bls381_witness_fq6_inverse_batch(bls381_ctx* c, const uint32_t* x, uint32_t* inv, uint8_t* flags, size_t n) {
  return witness_inverse(c, 1, 6, x, inv, flags, n);
}
int bls381_witness_fq2_inverse_batch(bls381_ctx* c, const uint32_t* x, uint32_t* inv, uint8_t* flags, size_t n) {
  return witness_inverse(c, 2, 2, x, inv, flags, n);
}
static int witness_sqrt(bls381_ctx* c, int op, size_t fp_per_elem, const uint32_t* x, const uint8_t* sgn, uint32_t* out, uint8_t* flags, size_t n) {
  if (!c || (n && (!x || !sgn || !out))) return BLS381_E_INVALID;
  if (n == 0) return BLS381_OK;
  const size_t bytes = n * fp_per_elem * 48;
  return with_staging(c, x, bytes, sgn, n, out, bytes, flags, n,
                      [&](const uint64_t* dx, const uint64_t* dsgn, uint64_t* o, uint8_t* f) -> int {
                        bls_launch::witness_sqrt(op, dx, (const uint8_t*)dsgn, o, f, n, c->block, c->stream);
                        HIP_TRY(c, hipGetLastError());
                        return (int)BLS381_OK;
                      });
}
int bls381_witness_fq_sqrt_batch(bls381_ctx* c, const uint32_t* x, const uint8_t* sgn, uint32_t* sqrt, uint8_t* flags, size_t n) {
  return witness_sqrt(c, 0, 1, x, sgn, sqrt, flags, n);
}
int bls381_witness_fq2_sqrt_batch(bls381_ctx* c, const uint32_t* x, const uint8_t* sgn, uint32_t* sqrt, uint8_t* flags, size_t n) {
  return witness_sqrt(c, 1, 2, x, sgn, sqrt, flags, n);
}

// Host-side coefficient-order helpers (pure permutations of 48-byte Fp slots, no device work):
// arkworks tree order c0.c0.c0, c0.c0.c1, c0.c1.c0, ... <-> MyFq12 flat order coeffs[2q + h + 6c]
// (src/fields/helpers.rs:22-76: coeffs[k] = w^k u^0 part, coeffs[k + 6] = w^k u^1 part).
void bls381_fq12_ark_to_myfq12(const uint64_t* in, uint64_t* out, size_t n) {
  for (size_t i = 0; i < n; ++i)
    for (int h = 0; h < 2; ++h)
      for (int q = 0; q < 3; ++q)
        for (int cc = 0; cc < 2; ++cc)
          std::memcpy(out + 72 * i + 6 * (2 * q + h + 6 * cc), in + 72 * i + 6 * (6 * h + 2 * q + cc), 48);
}
void bls381_fq12_myfq12_to_ark(const uint64_t* in, uint64_t* out, size_t n) {
  for (size_t i = 0; i < n; ++i)
    for (int h = 0; h < 2; ++h)
      for (int q = 0; q < 3; ++q)
        for (int cc = 0; cc < 2; ++cc)
          std::memcpy(out + 72 * i + 6 * (6 * h + 2 * q + cc), in + 72 * i + 6 * (2 * q + h + 6 * cc), 48);
}

int bls381_multi_miller_literal(bls381_ctx* c, uint64_t* out, size_t n) {
  if (!c || (n && !out)) return BLS381_E_INVALID;
  if (n == 0) return BLS381_OK;
  return with_staging(c, nullptr, 0, nullptr, 0, out, n * 576, nullptr, n,
                      [&](const uint64_t*, const uint64_t*, uint64_t* o, uint8_t*) -> int {
                        bls_launch::fill_one(o, n, c->io_form, c->stream);
                        HIP_TRY(c, hipGetLastError());
                        return (int)BLS381_OK;
                      });
}

// ------------------------------------------------------------------------------------------ several devices, one host process
// SURVEY 8e for a host that is ONE process (the reference's caller is a single Rust program): one context per device, the batch cut
// into contiguous ranges [g n / G, (g+1) n / G), every range through its device's host-pointer entry point (pinned-speed copies on
// the context's copy streams, kernels on its own stream) from its own host thread.  Elements are independent: nothing is exchanged.
struct bls381_multi {
  std::vector<bls381_ctx*> ctx;
};

int bls381_create_multi(bls381_multi** out, const int* device_ids, int n_devices) {
  if (!out) return BLS381_E_INVALID;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return BLS381_E_NO_DEVICE;
  bls381_multi* m = nullptr;
  try {                                        // std::vector growth may throw: nothing unwinds through the C ABI
    std::vector<int> ids;
    if (!device_ids || n_devices <= 0) { for (int i = 0; i < count; ++i) ids.push_back(i); }
    else ids.assign(device_ids, device_ids + n_devices);
    m = new bls381_multi();
    m->ctx.reserve(ids.size());
    for (int id : ids) {
      bls381_ctx* c = nullptr;
      const int rc = bls381_create(&c, id);
      if (rc != BLS381_OK) { bls381_destroy_multi(m); return rc; }
      m->ctx.push_back(c);
    }
  } catch (...) {
    bls381_destroy_multi(m);
    return BLS381_E_NOMEM;
  }
  *out = m;
  return BLS381_OK;
}

void bls381_destroy_multi(bls381_multi* m) {
  if (!m) return;
  for (bls381_ctx* c : m->ctx) bls381_destroy(c);
  delete m;
}

int bls381_multi_device_count(const bls381_multi* m) { return m ? (int)m->ctx.size() : 0; }

bls381_ctx* bls381_multi_context(bls381_multi* m, int index) {
  return (m && index >= 0 && index < (int)m->ctx.size()) ? m->ctx[index] : nullptr;
}

int bls381_multi_set_option(bls381_multi* m, int option, int value) {
  if (!m) return BLS381_E_INVALID;
  for (bls381_ctx* c : m->ctx) { const int rc = bls381_set_option(c, option, value); if (rc != BLS381_OK) return rc; }
  return BLS381_OK;
}

}  // extern "C"

namespace {
// run(ctx, first, count) on every device's contiguous range, one host thread per device; the first error wins
template <class Run>
int sharded(bls381_multi* m, size_t n, Run run) {
  const size_t G = m->ctx.size();
  // No exception leaves this function (the callers are extern "C"): allocation failures answer BLS381_E_NOMEM; a range whose
  // thread cannot be started (std::system_error) runs inline on the calling thread after the started ones are joined.
  try {
    std::vector<int> rc(G, BLS381_OK);
    std::vector<char> pending(G, 0);
    std::vector<std::thread> th;
    th.reserve(G);
    for (size_t g = 0; g < G; ++g) {
      const size_t lo = n * g / G, hi = n * (g + 1) / G;
      if (hi == lo) continue;
      if (G == 1) { rc[g] = run(m->ctx[g], lo, hi - lo); continue; }
      try {
        th.emplace_back([&rc, &run, m, g, lo, hi] { rc[g] = run(m->ctx[g], lo, hi - lo); });
      } catch (...) {
        pending[g] = 1;
      }
    }
    for (auto& t : th) t.join();
    for (size_t g = 0; g < G; ++g)
      if (pending[g]) rc[g] = run(m->ctx[g], n * g / G, n * (g + 1) / G - n * g / G);
    for (int r : rc) if (r != BLS381_OK) return r;
    return BLS381_OK;
  } catch (...) {
    return BLS381_E_NOMEM;
  }
}
}  // namespace

extern "C" {

int bls381_multi_pairing_batch_sharded(bls381_multi* m, const uint64_t* g1, const uint64_t* g2, size_t k, uint64_t* out, uint8_t* flags, size_t n) {
  if (!m || m->ctx.empty() || k < 1 || (n && (!g1 || !g2 || !out))) return BLS381_E_INVALID;
  return sharded(m, n, [&](bls381_ctx* c, size_t first, size_t cnt) {
    return bls381_multi_pairing_batch(c, g1 + 12 * k * first, g2 + 24 * k * first, k, out + 72 * first, flags ? flags + first : nullptr, cnt);
  });
}

int bls381_multi_miller_batch_sharded(bls381_multi* m, const uint64_t* g1, const uint64_t* g2, size_t k, uint64_t* out, uint8_t* flags, size_t n) {
  if (!m || m->ctx.empty() || k < 1 || (n && (!g1 || !g2 || !out))) return BLS381_E_INVALID;
  return sharded(m, n, [&](bls381_ctx* c, size_t first, size_t cnt) {
    return bls381_multi_miller_batch(c, g1 + 12 * k * first, g2 + 24 * k * first, k, out + 72 * first, flags ? flags + first : nullptr, cnt);
  });
}

int bls381_groth16_pairing_batch_sharded(bls381_multi* m, const uint64_t* g1, const uint64_t* g2_var, const uint64_t* prepared_fixed,
                                         uint64_t* out, uint8_t* flags, size_t n) {
  if (!m || m->ctx.empty() || (n && (!g1 || !g2_var || !prepared_fixed || !out))) return BLS381_E_INVALID;
  return sharded(m, n, [&](bls381_ctx* c, size_t first, size_t cnt) {
    return bls381_groth16_pairing_batch(c, g1 + 36 * first, g2_var + 24 * first, prepared_fixed, out + 72 * first, flags ? flags + first : nullptr, cnt);
  });
}

int bls381_miller_ref_opt_batch_sharded(bls381_multi* m, const uint64_t* g1, const uint64_t* g2, uint64_t* out, uint8_t* flags, size_t n) {
  if (!m || m->ctx.empty() || (n && (!g1 || !g2 || !out))) return BLS381_E_INVALID;
  return sharded(m, n, [&](bls381_ctx* c, size_t first, size_t cnt) {
    return bls381_miller_ref_opt_batch(c, g1 + 18 * first, g2 + 36 * first, out + 72 * first, flags ? flags + first : nullptr, cnt);
  });
}

}  // extern "C"
