// Replay of iteration bodies as HIP graphs -- built, bit-identical to the
// launches one by one, measured, and OFF by default.
//
// A Krylov iteration of this library is a fixed chain of dependent launches
// whose arguments do not change from one iteration to the next (the device
// decides about convergence: common.h `stopped`).  The solvers can capture an
// iteration body once (stream capture on the stream they are called with),
// keep the instantiated graph under a hash of everything that determines the
// body's launches, and replay it: FLOW_AMD_GRAPHS=1 / flow_graph_mode.
//
// The key is a hash of the VALUES that reach the kernels (the bytes of the
// operator / preconditioner structs, the vectors' addresses, the scalars), not
// of the structs' addresses: a caller that re-packs an operator into new
// buffers gets a new key, never a graph over freed memory.
//
// What was measured on MI355X / ROCm 7.2 (profiles/NOTES.md section 6, round 5):
// chains of EMPTY-ish kernels replay at 2.0-2.7 us per node against 3.5 us per
// launch, with 5 us of host time per chain (tools/micro/graph_launch.hip) --
// but from ~1 M doubles per kernel on a node costs 0.3-0.7 us MORE than the
// launch it replaces, and in the solvers that is what decides: the
// eighth-size proxy steps in 2.61 ms with every loop replayed (121 instead of
// 272 submissions per step) against 2.37-2.40 ms launched one by one, a
// 150 k-DoF channel in 1.90-1.94 against 1.83-1.86 ms, the 9.87 M-DoF
// workload in 9.88 against 9.85 ms.  The host was never the bottleneck of
// these loops (it runs ahead of the device between two read-backs); what a
// graph changes is the dispatch of each node, and that is not cheaper.
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>

#include "common.h"

namespace flow {

unsigned long long g_graph_replays = 0, g_graph_captures = 0,
                   g_graph_nodes = 0;

namespace {

struct Entry {
  hipGraphExec_t exec;
  int nodes;
  unsigned long long used;
};
std::unordered_map<unsigned long long, Entry> g_cache;
unsigned long long g_clock = 0;
int g_mode = -1;             // 0 off, 1 on wherever a solver asks, 2 auto
long long g_auto_rows = 0;
int g_sites = kReplayCg | kReplayGmres | kReplayMass;
bool g_broken = false;       // a capture failed: no further attempts
constexpr size_t kMaxEntries = 192;
constexpr int kThrashMisses = 32, kThrashPause = 400;
struct Thrash {
  int misses = 0, pause = 0;
};
Thrash g_thrash[8];
// the cache and its counters (a capture runs under the lock: two threads that
// solve on two streams take turns capturing, replays only look the graph up)
std::mutex g_lock;

void read_env() {
  if (g_mode >= 0) return;
  const char* e = getenv("FLOW_AMD_GRAPHS");
  if (!e || !*e) g_mode = 0;
  else if (!strcmp(e, "auto")) g_mode = 2;
  else g_mode = atoi(e) ? 1 : 0;
  const char* m = getenv("FLOW_AMD_GRAPH_SITES");
  if (m && *m) g_sites = atoi(m);
  const char* r = getenv("FLOW_AMD_GRAPH_ROWS");
  g_auto_rows = r && *r ? atoll(r) : kReplayAutoRows;
}

void drop_all() {
  if (g_cache.empty()) return;
  (void)hipDeviceSynchronize();
  for (auto& kv : g_cache) (void)hipGraphExecDestroy(kv.second.exec);
  g_cache.clear();
}

}  // namespace

bool replay_wanted(long long rows, int site) {
  read_env();
  if (g_broken || g_mode == 0 || !(g_sites & site)) return false;
  return g_mode == 1 || rows <= g_auto_rows;
}

unsigned long long g_site_captures[8] = {0, 0, 0, 0, 0, 0, 0, 0};

int replay_prepare(unsigned long long key, int site, hipStream_t st,
                   const std::function<int()>& body, hipGraphExec_t* exec,
                   int* nodes) {
  *exec = nullptr;
  *nodes = 0;
  std::lock_guard<std::mutex> guard(g_lock);
  Thrash& th = g_thrash[site & 7];
  auto it = g_cache.find(key);
  if (it != g_cache.end()) {
    it->second.used = ++g_clock;
    *exec = it->second.exec;
    *nodes = it->second.nodes;
    th.misses = 0;
    return FLOW_OK;
  }
  // a loop whose bodies are never seen twice (a caller that hands in fresh
  // vectors every time, a preconditioner rebuilt before every solve) only
  // pays for captures: after kThrashMisses in a row the loop launches
  // directly for the next kThrashPause requests, then tries again
  if (th.pause > 0) {
    --th.pause;
    return FLOW_OK;
  }
  if (++th.misses > kThrashMisses) {
    th.misses = 0;
    th.pause = kThrashPause;
    return FLOW_OK;
  }
  if (g_cache.size() >= kMaxEntries) {
    // (evict the older half -- behind a synchronisation: one of them may
    // still be executing)
    // (the WHOLE device: entries are shared by every stream and thread that
    // replays -- FLOW_AMD_FOLLOW_TORCH_STREAM, two solver threads --, a graph
    // may still be executing on another stream than the caller's)
    (void)hipStreamSynchronize(st);
    (void)hipDeviceSynchronize();
    std::vector<std::pair<unsigned long long, unsigned long long>> age;
    for (auto& kv : g_cache) age.push_back({kv.second.used, kv.first});
    std::sort(age.begin(), age.end());
    for (size_t i = 0; i < age.size() / 2; ++i) {
      (void)hipGraphExecDestroy(g_cache[age[i].second].exec);
      g_cache.erase(age[i].second);
    }
  }
  // capture: the body only ENQUEUES (no read-back, no synchronisation)
  const unsigned long long before = g_launches;
  if (hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed) != hipSuccess) {
    (void)hipGetLastError();
    g_broken = true;
    return FLOW_OK;         // the caller launches directly
  }
  const int rc = body();
  hipGraph_t graph = nullptr;
  const hipError_t e1 = hipStreamEndCapture(st, &graph);
  const int captured = static_cast<int>(g_launches - before);
  g_launches = before;      // (nothing was launched)
  if (rc) {
    if (graph) (void)hipGraphDestroy(graph);
    return rc;
  }
  if (e1 != hipSuccess || graph == nullptr) {
    (void)hipGetLastError();
    g_broken = true;
    return FLOW_OK;
  }
  hipGraphExec_t ge = nullptr;
  const hipError_t e2 = hipGraphInstantiate(&ge, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e2 != hipSuccess || ge == nullptr) {
    (void)hipGetLastError();
    g_broken = true;
    return FLOW_OK;
  }
  ++g_graph_captures;
  ++g_site_captures[site & 7];
  g_cache[key] = Entry{ge, captured, ++g_clock};
  *exec = ge;
  *nodes = captured;
  return FLOW_OK;
}

int replay_launch(hipGraphExec_t exec, int nodes, hipStream_t st) {
  FLOW_CHECK_HIP(hipGraphLaunch(exec, st));
  std::lock_guard<std::mutex> guard(g_lock);
  ++g_launches;             // one submission
  ++g_graph_replays;
  g_graph_nodes += static_cast<unsigned long long>(nodes);
  return FLOW_OK;
}

}  // namespace flow

using namespace flow;

extern "C" int flow_graph_stats(unsigned long long* stats_host) {
  FLOW_REQUIRE(stats_host != nullptr, "flow_graph_stats argument");
  std::lock_guard<std::mutex> guard(g_lock);
  stats_host[0] = g_cache.size();
  stats_host[1] = g_graph_captures;
  stats_host[2] = g_graph_replays;
  stats_host[3] = g_graph_nodes;
  stats_host[4] = g_site_captures[kReplayCg];
  stats_host[5] = g_site_captures[kReplayGmres];
  stats_host[6] = g_site_captures[kReplayMass];
  stats_host[7] = 0;
  return FLOW_OK;
}

extern "C" int flow_graph_mode(int mode, long long auto_rows) {
  // (bits 4.. of mode, when set: which loops -- 1 CG, 2 GMRES, 4 mass solver)
  const int sites = mode >> 4;
  mode &= 15;
  FLOW_REQUIRE(mode >= 0 && mode <= 2, "graph mode: 0 off, 1 on, 2 by size");
  std::lock_guard<std::mutex> guard(g_lock);
  read_env();
  g_mode = mode;
  if (sites) g_sites = sites;
  if (auto_rows >= 0) g_auto_rows = auto_rows;
  if (mode == 0) drop_all();
  for (Thrash& th : g_thrash) th = Thrash();
  g_broken = false;
  return FLOW_OK;
}
