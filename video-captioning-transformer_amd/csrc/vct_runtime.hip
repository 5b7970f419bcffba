// Host runtime of libvct_hip.so: recorded launch lists, cross-stream ordering and live kernel timing.
//
// Why: one training step is ~100 kernel launches on two HIP streams.  Issued from Python through ctypes the host
// needs ~1.6 ms per step for them (descriptor marshalling + planning + launch), which is the same order as the GPU
// time of the step, and the ORDER in which the host reaches the launches already shapes the overlap of the two
// streams.  hipGraph replay removes the host cost but serialises the two branches on this stack (measured -8 %).
// A launch list keeps eager semantics: it is the exact sequence of hipLaunchKernelGGL / hipEventRecord /
// hipStreamWaitEvent calls of one eager step, recorded once per shape configuration (vct::launch in vct_runtime.h
// captures kernel + geometry + argument copies) and re-issued from one C loop.
//
// replaces: nothing in the reference (its step is Python calling torch.nn modules, train.py:119-131); this is the
// executor under trainer.CaptionTrainer.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <functional>
#include <mutex>
#include <vector>
#include "../../include/vct_hip.h"
#include "vct_runtime.h"

namespace vct {

struct Cmd {
  int slot_a, slot_b;                                        // stream slots (b: second stream of a wait edge, else -1)
  std::function<void(hipStream_t, hipStream_t)> fn;
};
struct CmdList {
  std::vector<hipStream_t> streams;                          // slot -> stream recorded; slot 0 = the caller's main stream
  std::vector<Cmd> cmds;
  bool recording = false;
  int slot_of(hipStream_t s) {
    for (size_t i = 0; i < streams.size(); i++)
      if (streams[i] == s) return (int)i;
    streams.push_back(s);
    return (int)streams.size() - 1;
  }
};

thread_local CmdList* g_rec = nullptr;
static thread_local int g_replay_err = 0;          // first error noted by a closure of the replay in progress
void replay_note_error(int code) { if (code != 0 && g_replay_err == 0) g_replay_err = code; }

void rec_push(hipStream_t st, std::function<void(hipStream_t)>&& fn) {
  CmdList* l = g_rec;
  Cmd c;
  c.slot_a = l->slot_of(st);
  c.slot_b = -1;
  c.fn = [f = std::move(fn)](hipStream_t a, hipStream_t) { f(a); };
  l->cmds.push_back(std::move(c));
}

// ---- events for cross-stream edges ------------------------------------------------------------------------------
// ids 0..63 are the caller's named sync points, 64..255 rotate under vct_stream_wait.  An event may be re-recorded while
// an earlier wait on it is still queued: a wait binds to the most recent record issued before it (host order), and one
// host thread issues both.
constexpr int N_EVENTS = 256, N_NAMED = 64;
static hipEvent_t g_events[N_EVENTS];
static bool g_event_made[N_EVENTS];
static int g_rot = 0;
static std::mutex g_mu;

static hipEvent_t event_of(int id) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_event_made[id]) {
    if (hipEventCreateWithFlags(&g_events[id], hipEventDisableTiming) != hipSuccess) return nullptr;
    g_event_made[id] = true;
  }
  return g_events[id];
}

static int do_record(int id, hipStream_t st) {
  hipEvent_t e = event_of(id);
  if (e == nullptr) return (int)hipErrorOutOfMemory;
  if (g_rec != nullptr) {
    CmdList* l = g_rec;
    Cmd c; c.slot_a = l->slot_of(st); c.slot_b = -1;
    c.fn = [e](hipStream_t a, hipStream_t) { const hipError_t r = hipEventRecord(e, a); if (r != hipSuccess) replay_note_error((int)r); };
    l->cmds.push_back(std::move(c));
    return VCT_OK;
  }
  const hipError_t r = hipEventRecord(e, st);
  return r == hipSuccess ? VCT_OK : (int)r;
}
static int do_wait(int id, hipStream_t st) {
  hipEvent_t e = event_of(id);
  if (e == nullptr) return (int)hipErrorOutOfMemory;
  if (g_rec != nullptr) {
    CmdList* l = g_rec;
    Cmd c; c.slot_a = l->slot_of(st); c.slot_b = -1;
    c.fn = [e](hipStream_t a, hipStream_t) { const hipError_t r = hipStreamWaitEvent(a, e, 0); if (r != hipSuccess) replay_note_error((int)r); };
    l->cmds.push_back(std::move(c));
    return VCT_OK;
  }
  const hipError_t r = hipStreamWaitEvent(st, e, 0);
  return r == hipSuccess ? VCT_OK : (int)r;
}

// ---- live timing taps ---------------------------------------------------------------------------------------------
constexpr int N_TAGS = 24;
struct TapPair { hipEvent_t a, b; };
static std::vector<TapPair> g_tap_pool[N_TAGS];       // every pair ever created for the tag
static int g_tap_used[N_TAGS];                        // pairs handed out since the last collect
static bool g_tap_on = false;

static TapPair* tap_pair(int tag, bool fresh) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (fresh) {
    if (g_tap_used[tag] == (int)g_tap_pool[tag].size()) {
      TapPair p;
      if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return nullptr;
      g_tap_pool[tag].push_back(p);
    }
    g_tap_used[tag]++;
  }
  if (g_tap_used[tag] == 0) return nullptr;
  return &g_tap_pool[tag][g_tap_used[tag] - 1];
}
static void tap_exec(int tag, int phase, hipStream_t st) {
  if (!g_tap_on) return;
  TapPair* p = tap_pair(tag, phase == 0);
  if (p != nullptr) (void)hipEventRecord(phase == 0 ? p->a : p->b, st);
}

}  // namespace vct
using namespace vct;

extern "C" int vct_cmdlist_create(void** out) {
  if (out == nullptr) return VCT_E_ARG;
  *out = new (std::nothrow) CmdList();
  return *out ? VCT_OK : (int)hipErrorOutOfMemory;
}
extern "C" int vct_cmdlist_destroy(void* list) {
  if (list == nullptr) return VCT_E_ARG;
  if (g_rec == list) g_rec = nullptr;
  delete reinterpret_cast<CmdList*>(list);
  return VCT_OK;
}
extern "C" int vct_cmdlist_begin(void* list, void* main_stream) {
  if (list == nullptr || g_rec != nullptr) return VCT_E_ARG;      // one recording per thread at a time
  CmdList* l = reinterpret_cast<CmdList*>(list);
  l->cmds.clear();
  l->streams.clear();
  l->streams.push_back((hipStream_t)main_stream);
  l->recording = true;
  g_rec = l;
  return VCT_OK;
}
extern "C" int vct_cmdlist_end(void* list) {
  if (list == nullptr || g_rec != list) return VCT_E_ARG;
  g_rec->recording = false;
  g_rec = nullptr;
  return VCT_OK;
}
extern "C" int vct_cmdlist_size(void* list) {
  return list == nullptr ? VCT_E_ARG : (int)reinterpret_cast<CmdList*>(list)->cmds.size();
}
extern "C" int vct_cmdlist_streams(void* list) {
  return list == nullptr ? VCT_E_ARG : (int)reinterpret_cast<CmdList*>(list)->streams.size();
}
extern "C" int vct_cmdlist_replay(void* list, void* main_stream) {
  if (list == nullptr) return VCT_E_ARG;
  CmdList* l = reinterpret_cast<CmdList*>(list);
  if (l->recording || g_rec != nullptr) return VCT_E_ARG;
  hipStream_t map[16];
  const int ns = (int)l->streams.size();
  if (ns > 16) return VCT_E_SHAPE;
  for (int i = 0; i < ns; i++) map[i] = l->streams[i];
  map[0] = (hipStream_t)main_stream;
  if (l->cmds.empty()) return VCT_OK;
  g_replay_err = 0;
  for (const Cmd& c : l->cmds) c.fn(map[c.slot_a], c.slot_b >= 0 ? map[c.slot_b] : nullptr);
  const hipError_t e = hipGetLastError();
  if (g_replay_err != 0) return g_replay_err;      // first failed collective / event call of this replay (sticky)
  return e == hipSuccess ? VCT_OK : (int)e;
}

extern "C" int vct_cmdlist_inject_status(int status, void* stream) {
  if (g_rec == nullptr) return status;
  rec_push((hipStream_t)stream, [status](hipStream_t) { replay_note_error(status); });
  return VCT_OK;
}

extern "C" int vct_cmdlist_host_call(int (*fn)(void*), void* arg, void* stream) {
  if (fn == nullptr) return VCT_E_ARG;
  auto run = [fn, arg](hipStream_t s) -> int {
    const hipError_t e = hipStreamSynchronize(s);
    if (e != hipSuccess) return (int)e;
    return fn(arg);
  };
  if (g_rec != nullptr) {
    rec_push((hipStream_t)stream, [run](hipStream_t s) { replay_note_error(run(s)); });
    return VCT_OK;
  }
  return run((hipStream_t)stream);
}

extern "C" int vct_sync_record(int id, void* stream) {
  if (id < 0 || id >= N_NAMED) return VCT_E_ARG;
  return do_record(id, (hipStream_t)stream);
}
extern "C" int vct_sync_wait(int id, void* stream) {
  if (id < 0 || id >= N_NAMED) return VCT_E_ARG;
  return do_wait(id, (hipStream_t)stream);
}
extern "C" int vct_stream_wait(void* waiter, void* signal) {
  if (waiter == signal) return VCT_OK;
  int id;
  { std::lock_guard<std::mutex> lk(g_mu); id = N_NAMED + (g_rot++ % (N_EVENTS - N_NAMED)); }
  const int r = do_record(id, (hipStream_t)signal);
  if (r != VCT_OK) return r;
  return do_wait(id, (hipStream_t)waiter);
}

// ---- CU-masked streams ------------------------------------------------------------------------------------------------
extern "C" int vct_stream_create_masked(const uint32_t* cu_mask, int words, void** out_stream) {
  if (out_stream == nullptr || (words > 0 && cu_mask == nullptr) || words < 0) return VCT_E_ARG;
  hipStream_t s = nullptr;
  hipError_t e = words == 0 ? hipStreamCreateWithFlags(&s, hipStreamNonBlocking)
                            : hipExtStreamCreateWithCUMask(&s, (uint32_t)words, cu_mask);
  if (e != hipSuccess) return (int)e;
  *out_stream = (void*)s;
  return VCT_OK;
}
extern "C" int vct_stream_destroy(void* stream) {
  if (stream == nullptr) return VCT_E_ARG;
  const hipError_t e = hipStreamDestroy((hipStream_t)stream);
  return e == hipSuccess ? VCT_OK : (int)e;
}

extern "C" int vct_tap_enable(int on) { g_tap_on = on != 0; return VCT_OK; }
extern "C" int vct_tap(int tag, int phase, void* stream) {
  if (tag < 0 || tag >= N_TAGS || (phase != 0 && phase != 1)) return VCT_E_ARG;
  if (!g_tap_on) return VCT_OK;
  hipStream_t st = (hipStream_t)stream;
  if (g_rec != nullptr) {
    rec_push(st, [tag, phase](hipStream_t s) { tap_exec(tag, phase, s); });
    return VCT_OK;
  }
  tap_exec(tag, phase, st);
  return VCT_OK;
}
extern "C" int vct_tap_collect(int tag, float* ms_out, int cap) {
  if (tag < 0 || tag >= N_TAGS || (ms_out == nullptr && cap > 0)) return VCT_E_ARG;
  int n;
  { std::lock_guard<std::mutex> lk(g_mu); n = g_tap_used[tag]; g_tap_used[tag] = 0; }
  int k = 0;
  for (int i = 0; i < n && k < cap; i++) {
    TapPair& p = g_tap_pool[tag][i];
    if (hipEventSynchronize(p.b) != hipSuccess) continue;
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) ms_out[k++] = ms;
  }
  return k;
}
