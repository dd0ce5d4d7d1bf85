// Host-side symbolic analysis (see gn_symbolic.h).  Plain C++17, no GPU calls, so it is
// unit-testable on a CPU-only machine through cgmr_gn_symbolic_info().
#include "gn_symbolic.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <sched.h>
#include <pthread.h>
#include <thread>
#include <unistd.h>

namespace cgmr {
namespace {

double now_s() {
  using namespace std::chrono;
  return duration_cast<duration<double>>(steady_clock::now().time_since_epoch()).count();
}

constexpr int kRangeDepth = 3;        // ND depth whose halves become the parallel units of the border computation

// A few persistent helper threads: the analysis forks a dozen short parallel sections per call, and creating a
// thread for each costs more than most of them run.  Idle helpers spin briefly (the next section usually follows
// within microseconds), then sleep.
class HelperPool {
 public:
  struct Job {
    std::function<void()> fn;
    std::atomic<int> done{0};
    const void* owner = nullptr;         // the submitting thread (help_until() only takes its own thread's queued jobs)
  };
  static const void* self() { static thread_local char tag; return &tag; }
  explicit HelperPool(int nhelpers) : owner_(getpid()) {
    for (int i = 0; i < nhelpers; i++) threads_.emplace_back([this] { loop(); });
    pin_near_caller();
  }
  ~HelperPool() {
    if (getpid() != owner_) {                      // a forked child inherits the object but not the threads
      for (auto& t : threads_) t.detach();
      return;
    }
    { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
    stop_flag_.store(true, std::memory_order_relaxed);
    cv_.notify_all();
    for (auto& t : threads_) t.join();
  }
  int helpers() const { return (int)threads_.size(); }
  int moves() { std::lock_guard<std::mutex> lk(state_mu_); return moves_; }
  void account_caller(long long wall_ns, long long cpu_ns) {
    wall_ns_.fetch_add(wall_ns, std::memory_order_relaxed);
    cpu_ns_.fetch_add(cpu_ns, std::memory_order_relaxed);
  }
  static long long now_ns(clockid_t id) { return clock_ns(id); }
  // The GPU boxes are shared: when another process keeps the helpers' cores busy (every process of this library started
  // from the same launcher CPU would pick the same cache group), a helper's job takes its wall time in time slices and an
  // analysis on 8 threads costs what it costs on one (seen as one bench run in five at 3.5-5.4 ms of analysis instead of
  // 2.3).  The helpers time themselves while they spin or work, the caller its analysis -- elapsed against thread CPU time;
  // after four analyses in a row in which they were off their cores for more than a third of that time the pool moves to another cache group of the same
  // socket (the caller's home with it).  Called by analyze() once the caller is back on its own affinity mask.
  void rebalance() {
    // opt-in (CGMR_HOST_MOVE=1): a library inside somebody else's process does not move threads around on its own account
    static const bool on = getenv("CGMR_HOST_MOVE") && atoi(getenv("CGMR_HOST_MOVE")) != 0;
    const long long w = wall_ns_.exchange(0, std::memory_order_relaxed), c = cpu_ns_.exchange(0, std::memory_order_relaxed);
    if (!on || getpid() != owner_) return;
    std::lock_guard<std::mutex> lk(state_mu_);                    // (several caller threads analyse: a context per robot, a thread each)
    if (home_.empty() || group_cur_ < 0 || group_firsts_.size() < 2) return;
    if (w < 200000) return;                                       // (less than 0.2 ms of helper work: no verdict)
    if (3 * c >= 2 * w) { strikes_ = 0; return; }
    if (++strikes_ < (4 << std::min(moves_, 5))) return;           // (every move makes the next one harder: a host that is busy everywhere is not fled from)
    strikes_ = 0;
    if (home_busy_.exchange(true, std::memory_order_acquire)) return;      // (another caller is at home right now: next time)
    const int n = (int)group_firsts_.size();
    // far away, by an odd number of groups (every group comes up; ranks placed on even groups are not met), and not by the
    // number a neighbour that started on the same group moves by
    const int step = n > 4 ? ((n / 2 + 1) | 1) + 2 * (int)(getpid() % 3) : 1 + (int)(getpid() % 2);
    for (int k = 1; k < n; k++) {
      const int g = (group_cur_ + k * step) % n;
      if (g == group_cur_) continue;
      if (pin_around(group_firsts_[g])) { group_cur_ = g; moves_++; break; }
    }
    home_busy_.store(false, std::memory_order_release);
  }
  int home_cpu() { std::lock_guard<std::mutex> lk(state_mu_); return home_.empty() ? -1 : home_[0]; }
  // run `job` on a helper if one is idle, else right here; wait() returns when it is done
  void run(Job& job) {
    job.done.store(0, std::memory_order_relaxed);
    job.owner = self();
    bool queued = false;
    if (getpid() == owner_) {
      std::lock_guard<std::mutex> lk(mu_);
      if (busy_ + (int)queue_.size() < (int)threads_.size()) { queue_.push_back(&job); queued = true; }
    }
    if (!queued) { job.fn(); job.done.store(1, std::memory_order_release); return; }
    pending_.fetch_add(1, std::memory_order_release);
    cv_.notify_one();
  }
  // Returns when `job` is done.  While it is not, the waiting thread takes queued jobs of its OWN that nobody has started yet
  // and runs them itself: a helper whose core is busy with somebody else's work -- the GPU boxes are shared -- then costs
  // its share of the section, not a scheduler time slice.  (Jobs of other caller threads are left alone: taking another
  // robot's long job would hold this caller up behind work it never asked for.)
  static void wait(Job& job);
  void help_until(Job& job) {
    static const bool steal = !(getenv("CGMR_HOST_STEAL") && atoi(getenv("CGMR_HOST_STEAL")) == 0);
    while (!job.done.load(std::memory_order_acquire)) {
      Job* other = nullptr;
      if (steal && pending_.load(std::memory_order_acquire) > 0 && getpid() == owner_) {
        std::lock_guard<std::mutex> lk(mu_);
        const void* me = self();
        for (size_t k = 0; k < queue_.size(); k++)
          if (queue_[k]->owner == me) { other = queue_[k]; queue_.erase(queue_.begin() + k); pending_.fetch_sub(1); break; }
      }
      if (other) {
        other->fn();
        other->done.store(1, std::memory_order_release);
      } else {
        std::this_thread::yield();
      }
    }
  }

 private:
  // The helpers work on the caller's arrays (adjacency, queue slices, the permutation) a few hundred microseconds at a time:
  // on a two-socket / many-CCX host a helper that wakes up far from the caller pays for every line twice.  Each helper is
  // pinned to its own core among those that share the last-level cache with the CPU the pool is created from (read from
  // sysfs; nothing happens when that fails, when CGMR_HOST_PIN=0, or when the process's affinity mask excludes the cores).
  static int package_of(int cpu) {
    std::vector<int> v;
    return read_list("/sys/devices/system/cpu/cpu" + std::to_string(cpu) + "/topology/physical_package_id", v) ? v[0] : -1;
  }
  static bool read_list(const std::string& path, std::vector<int>& out) {      // a sysfs CPU list: "0-7,128-135"
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return false;
    char buf[4096];
    const bool ok = fgets(buf, sizeof buf, f) != nullptr;
    fclose(f);
    if (!ok) return false;
    for (char* q = buf; *q;) {
      char* end;
      long a = strtol(q, &end, 10);
      if (end == q) break;
      long b = a;
      if (*end == '-') { q = end + 1; b = strtol(q, &end, 10); }
      for (long c = a; c <= b && c < 4096; c++) out.push_back((int)c);
      q = (*end == ',') ? end + 1 : end;
      if (*end != ',') break;
    }
    return !out.empty();
  }
  void pin_near_caller() {
    static const bool off = getenv("CGMR_HOST_PIN") && atoi(getenv("CGMR_HOST_PIN")) == 0;
    if (off || threads_.empty() || getpid() != owner_) return;
    int me = sched_getcpu();
    if (me < 0) return;
    const std::string base = "/sys/devices/system/cpu/cpu";
    cpu_set_t allowed;
    CPU_ZERO(&allowed);
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return;
    // Several ranks on one node (one process per GPU, LOCAL_RANK / LOCAL_WORLD_SIZE set by the launcher): the ranks' callers
    // may well sit in the same cache group when their pools are created (they were forked from one parent), and two pools
    // pinned to the same eight cores would share them for good.  Rank k of n takes the (k * groups / n)-th cache group of
    // the machine instead of the one it happens to run in.
    {
      const char* lr = getenv("LOCAL_RANK");
      const char* lw = getenv("LOCAL_WORLD_SIZE") ? getenv("LOCAL_WORLD_SIZE") : getenv("WORLD_SIZE");   // (one node)
      const int k = lr ? atoi(lr) : 0, n = lw ? atoi(lw) : 1;
      if (lr && n > 1 && k >= 0 && k < n) {
        std::vector<int> firsts;                                  // first allowed CPU of every cache group, in CPU order
        std::vector<uint8_t> seen(4096, 0);
        const long ncpu = std::min(4096L, sysconf(_SC_NPROCESSORS_CONF));
        for (int c = 0; c < ncpu; c++) {
          if (seen[c]) continue;
          std::vector<int> grp;
          if (!read_list(base + std::to_string(c) + "/cache/index3/shared_cpu_list", grp)) { seen[c] = 1; continue; }
          int first = -1;
          for (int q : grp) { seen[q] = 1; if (first < 0 && CPU_ISSET(q, &allowed)) first = q; }
          if (first >= 0) firsts.push_back(first);
        }
        if ((int)firsts.size() >= n) me = firsts[(size_t)k * firsts.size() / n];
      }
    }
    // the cache groups of the machine (first allowed CPU of each), for a later move of the pool (rebalance())
    {
      std::vector<uint8_t> seen(4096, 0);
      const long ncpu = std::min(4096L, sysconf(_SC_NPROCESSORS_CONF));
      for (int c = 0; c < ncpu; c++) {
        if (seen[c]) continue;
        std::vector<int> grp;
        if (!read_list(base + std::to_string(c) + "/cache/index3/shared_cpu_list", grp)) { seen[c] = 1; continue; }
        int first = -1;
        bool mine = false;
        for (int q : grp) { seen[q] = 1; if (first < 0 && CPU_ISSET(q, &allowed)) first = q; mine = mine || q == me; }
        if (first < 0) continue;
        // only groups of the caller's own socket: its arrays, the pinned staging buffers and the GPU's host memory live there
        // (a pool that moved to the other socket analysed in 5.5-8.4 ms instead of 2.3)
        if (!mine && package_of(first) != package_of(me)) continue;
        if (mine) group_cur_ = (int)group_firsts_.size();
        group_firsts_.push_back(mine ? me : first);
      }
    }
    (void)pin_around(me);
  }

  // helpers onto the cores that share the last-level cache with CPU `me` (one per physical core, `me`'s own core left to
  // the caller); false and nothing changed if that group cannot take them
  bool pin_around(int me) {
    const std::string base = "/sys/devices/system/cpu/cpu";
    cpu_set_t allowed;
    CPU_ZERO(&allowed);
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return false;
    std::vector<int> l3;
    if (!read_list(base + std::to_string(me) + "/cache/index3/shared_cpu_list", l3)) return false;
    // one CPU per physical core (the first hardware thread listed for it), the caller's own core left to the caller
    std::vector<int> sib_me;
    read_list(base + std::to_string(me) + "/topology/thread_siblings_list", sib_me);
    std::vector<int> picks;
    std::vector<uint8_t> taken(4096, 0);
    for (int c : sib_me) taken[c] = 1;
    for (int c : l3) {
      if (taken[c] || !CPU_ISSET(c, &allowed)) continue;
      std::vector<int> sib;
      if (!read_list(base + std::to_string(c) + "/topology/thread_siblings_list", sib)) sib.assign(1, c);
      for (int q : sib) taken[q] = 1;
      picks.push_back(c);
    }
    if ((int)picks.size() < (int)threads_.size()) return false;   // fewer cores behind this cache than helpers: leave the scheduler alone
    bool all = true;
    static const int pin_mode = getenv("CGMR_HOST_PIN") ? atoi(getenv("CGMR_HOST_PIN")) : 1;
    cpu_set_t whole;                                               // CGMR_HOST_PIN=2: every helper anywhere in the group but on the caller's core
    CPU_ZERO(&whole);
    for (int c : l3) {
      bool mine = false;
      for (int q : sib_me) mine = mine || q == c;
      if (!mine && c != me && CPU_ISSET(c, &allowed)) CPU_SET(c, &whole);
    }
    for (size_t i = 0; i < threads_.size(); i++) {
      cpu_set_t one;
      CPU_ZERO(&one);
      CPU_SET(picks[i], &one);
      all = all && pthread_setaffinity_np(threads_[i].native_handle(), sizeof one, pin_mode == 2 ? &whole : &one) == 0;
    }
    if (all) home_ = sib_me.empty() ? std::vector<int>(1, me) : sib_me;     // (callers hold state_mu_ or are the constructor)
    return all;
  }

 public:
  // For the duration of an analysis the calling thread is held on the core the helpers were placed around (its "home": the
  // CPU the pool was created from, one core of that cache group that no helper uses, with its hardware-thread sibling).  A
  // caller that blocks between solves (a stream synchronisation, a collective) is woken up wherever the scheduler likes;
  // letting the seven helpers follow it across cache groups, or leaving it on a core where a pinned helper spins, made the
  // analysis 2.4 or 7.6 ms from one process to the next.  The caller's affinity mask is restored when the scope ends.
  class CallerAtHome {
   public:
    explicit CallerAtHome(HelperPool& p) {
      // opt-in (CGMR_HOST_PIN_CALLER=1): by default the library never touches the affinity of a thread it does not own -- a
      // ROS node that links it has its own ideas about where its threads run
      static const bool on = getenv("CGMR_HOST_PIN_CALLER") && atoi(getenv("CGMR_HOST_PIN_CALLER")) != 0;
      if (!on) return;
      // one caller at a time: a second thread of the process that analyses meanwhile (a context per robot, a thread each)
      // stays where it is instead of queueing up for the same core
      if (p.home_busy_.exchange(true, std::memory_order_acquire)) return;
      pool_ = &p;
      std::vector<int> home_cpus;
      { std::lock_guard<std::mutex> lk(p.state_mu_); home_cpus = p.home_; }   // (read under the flag and the lock: a move rewrites it)
      if (home_cpus.empty()) return;
      if (sched_getaffinity(0, sizeof saved_, &saved_) != 0) return;
      cpu_set_t home;
      CPU_ZERO(&home);
      bool any = false;
      for (int c : home_cpus) if (CPU_ISSET(c, &saved_)) { CPU_SET(c, &home); any = true; }
      if (!any) return;                                          // (the caller may not run there: leave it alone)
      active_ = sched_setaffinity(0, sizeof home, &home) == 0;
    }
    ~CallerAtHome() {
      if (active_) (void)sched_setaffinity(0, sizeof saved_, &saved_);
      if (pool_) pool_->home_busy_.store(false, std::memory_order_release);
    }
    CallerAtHome(const CallerAtHome&) = delete;
    CallerAtHome& operator=(const CallerAtHome&) = delete;
   private:
    cpu_set_t saved_;
    bool active_ = false;
    HelperPool* pool_ = nullptr;
  };

 private:
  void loop() {
    // awake = spinning for work or working: the thread wants its core all of that time, so elapsed time beyond its CPU time
    // is time somebody else had the core (rebalance())
    long long w0 = clock_ns(CLOCK_MONOTONIC), c0 = clock_ns(CLOCK_THREAD_CPUTIME_ID);
    auto account = [&] {
      const long long w1 = clock_ns(CLOCK_MONOTONIC), c1 = clock_ns(CLOCK_THREAD_CPUTIME_ID);
      wall_ns_.fetch_add(w1 - w0, std::memory_order_relaxed);
      cpu_ns_.fetch_add(c1 - c0, std::memory_order_relaxed);
      w0 = w1; c0 = c1;
    };
    for (;;) {
      Job* job = nullptr;
      // Spin on the pending counter before sleeping on the condition variable (CGMR_HOST_SPIN_US after the last job, default
      // 200 us: the sections of one analysis follow each other within microseconds).  A dedicated solve loop can ask for as
      // long as it takes to come back with the next analysis (the bench sets 10 ms: a helper that sleeps through the 5 ms of
      // device work between two analyses pays a futex wake-up each time, and on a shared host its core has been given to
      // somebody else in the meantime) -- not the default, because 8 ranks x 7 helpers would then never sleep.
      static const long long spin_ns = 1000LL * (getenv("CGMR_HOST_SPIN_US") ? std::max(0, atoi(getenv("CGMR_HOST_SPIN_US"))) : 200);
      const long long spin_until = clock_ns(CLOCK_MONOTONIC) + spin_ns;
      for (unsigned spin = 0; !job; spin++) {
        if (pending_.load(std::memory_order_acquire) > 0) {
          std::lock_guard<std::mutex> lk(mu_);
          if (!queue_.empty()) { job = queue_.front(); queue_.erase(queue_.begin()); busy_++; pending_.fetch_sub(1); }
        } else {
          __builtin_ia32_pause();
          if ((spin & 255) == 255 && (stop_flag_.load(std::memory_order_relaxed) || clock_ns(CLOCK_MONOTONIC) >= spin_until)) break;
        }
      }
      if (!job) {
        account();
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [this] { return stop_ || !queue_.empty(); });
        if (stop_) return;
        job = queue_.front(); queue_.erase(queue_.begin()); busy_++; pending_.fetch_sub(1);
        w0 = clock_ns(CLOCK_MONOTONIC); c0 = clock_ns(CLOCK_THREAD_CPUTIME_ID);
      }
      job->fn();
      account();
      { std::lock_guard<std::mutex> lk(mu_); busy_--; }
      job->done.store(1, std::memory_order_release);
    }
  }
  static long long clock_ns(clockid_t id) {
    timespec ts;
    clock_gettime(id, &ts);
    return (long long)ts.tv_sec * 1000000000LL + ts.tv_nsec;
  }
  std::vector<std::thread> threads_;
  std::vector<int> home_;                // the caller's core while it analyses (empty: the helpers are not pinned)
  std::atomic<bool> home_busy_{false};   // a caller is held there right now
  std::vector<int> group_firsts_;        // a CPU of every last-level-cache group the process may use (the pool's own: the home CPU)
  int group_cur_ = -1, strikes_ = 0, moves_ = 0;
  std::atomic<long long> wall_ns_{0}, cpu_ns_{0};   // the helpers' awake time (and the callers' analyses) since the last look: elapsed against CPU time
  std::vector<Job*> queue_;
  std::mutex mu_;
  std::mutex state_mu_;                  // home_, group_cur_, strikes_, moves_ (several caller threads analyse and rebalance)
  std::condition_variable cv_;
  std::atomic<int> pending_{0};
  int busy_ = 0;
  bool stop_ = false;
  std::atomic<bool> stop_flag_{false};
  pid_t owner_;
};

int host_threads();
HelperPool& pool() {
  static HelperPool p(host_threads() - 1);
  return p;
}
void HelperPool::wait(Job& job) { pool().help_until(job); }

// ------------------------------------------------------------------ nested dissection
// Recursive bisection by breadth-first level structures: pick the level that splits the
// subset most evenly among the thin ones, keep only its vertices that touch the next level
// as the separator.  Output: `order` (position -> block index) with subsets laid out as
// [part A][part B][separator], and `panel_start`, the positions at which a new front begins
// (leaves and separators are cut into runs of at most kPanelW columns).
struct NDCtx {
  const std::vector<int32_t>& ap;     // adjacency CSR (block indices)
  const std::vector<int32_t>& ai;
  std::vector<int32_t>& order;        // in/out working permutation
  std::vector<uint8_t>& pstart;       // pstart[pos] = 1 if a front begins at position pos
  std::mutex range_mu;
  std::vector<std::pair<int, int>> subtree_ranges;   // position ranges of the halves of the nodes at depth kRangeDepth:
                                                     // complete subtrees, independent of each other
  struct VState { int32_t label, dist; };   // region id a vertex currently belongs to / BFS depth: one cache line per visit
  std::vector<VState> vs;             // indexed by vertex
  std::vector<int32_t> queue;         // BFS order, indexed by *position*: a call only touches [begin, end)
  std::vector<int32_t> tmp;           // partition scratch, indexed by position
  std::atomic<int> next_label{1};
  int max_par_depth = 0;              // recursion levels that fork a thread for one half
  // the dissection tree, recorded for the incremental analysis of a graph that grows (sizes only: positions follow from
  // the root's, children may trade places after they were recorded)
  struct Rec { int32_t a = -1, b = -1, n = 0, ssz = 0; };
  std::vector<Rec> recs;              // preallocated: node ids are handed out by an atomic counter (the halves run in parallel)
  std::atomic<int> n_recs{0};
  int new_rec(int a, int b, int n, int ssz) {
    const int id = n_recs.fetch_add(1);
    if (id < (int)recs.size()) { recs[id].a = a; recs[id].b = b; recs[id].n = n; recs[id].ssz = ssz; }
    return id;
  }
};

// What a parent separator needs to know about an ordered range: the size of its last panel (the one the separator,
// placed right behind the range, could be merged with) and an estimate of the height of its elimination subtree in
// panels.
struct NDRange {
  int last = 0;
  int height = 0;
};

// cut [begin, end) into runs of `run` columns, the remainder last
NDRange emit_panels(NDCtx& C, int begin, int end, int run = kPanelW) {
  NDRange r;
  for (int p = begin; p < end; p += run) { C.pstart[p] = 1; r.height++; }
  r.last = end > begin ? (end - begin - 1) % run + 1 : 0;
  return r;
}

// BFS restricted to vertices with label == id; returns number reached, fills dist (by vertex)
// and q[0..reached) in visiting order (q = the caller's slice of C.queue).  The queue is filled level by level:
// with `lvl` (nullable, the caller's slice of C.tmp, zeroed here as far as it is used) lvl[d] = vertices at depth d.
int bfs(NDCtx& C, int32_t* q, int root, int id, int visited_id, int32_t* lvl = nullptr, NDCtx::VState* vs = nullptr) {
  if (!vs) vs = C.vs.data();                           // (a private copy of the states: the root's candidate sweeps run side by side)
  int qh = 0, qt = 0;
  q[qt++] = root;
  vs[root].dist = 0;
  vs[root].label = visited_id;
  int top = 0;                                          // deepest level counted so far
  if (lvl) lvl[0] = 1;
  // (a branch-free visit -- conditional moves, a sink for the stores not taken -- was measured 25 % slower)
  while (qh < qt) {
    int u = q[qh++];
    int du = vs[u].dist;
    for (int p = C.ap[u]; p < C.ap[u + 1]; p++) {
      int w = C.ai[p];
      NDCtx::VState& W = vs[w];
      if (W.label != id) continue;
      W.label = visited_id;
      W.dist = du + 1;
      if (lvl) { if (du + 1 > top) { top = du + 1; lvl[top] = 0; } lvl[du + 1]++; }
      q[qt++] = w;
    }
  }
  return qt;
}

// Orders [begin, end) and marks its panels.
// `start`: a vertex of the range known to lie at one end of it (the parent's sweep began or ended there), or -1.
NDRange nd(NDCtx& C, int begin, int end, int depth, int start = -1, int* node_out = nullptr) {
  int n = end - begin;
  static const bool nd_trace = getenv("CGMR_SYM_TRACE") != nullptr;
  const double t_in = nd_trace ? now_s() : 0;
  int node_sink = -1;
  int& node = node_out ? *node_out : node_sink;
  node = -1;
  if (n <= 0) return NDRange();
  if (n <= kPanelW) { node = C.new_rec(-1, -1, n, n); return emit_panels(C, begin, end); }
  int32_t* Q = C.queue.data() + begin;                  // this call's slice of the BFS queue
  int id = C.next_label.fetch_add(3);
  for (int p = begin; p < end; p++) C.vs[C.order[p]].label = id;
  int32_t* LV = C.tmp.data() + begin;                   // level counts of the structuring sweep (the slice is free until the partition)
  // first sweep: connectivity + a far vertex; skipped when the parent's sweep already left one (then the second
  // sweep doubles as the connectivity check)
  int vis1 = id + 1, vis2 = id + 2;
  static const bool reuse_start = !(getenv("CGMR_ND_REUSE_START") && atoi(getenv("CGMR_ND_REUSE_START")) == 0);
  const bool have_start = reuse_start && start >= 0;
  const double t_l0 = nd_trace ? now_s() : 0;
  // The root (round 6): instead of a sweep for a far vertex and a second one from there -- two whole-graph sweeps one behind
  // the other on one thread, 0.38 ms at C2's size, while the others idle --, K = 4 structuring sweeps side by side from the
  // vertices at 0, 1/3, 2/3 and the end of the range (the first and the last pose among them), the deepest level structure
  // kept (ties: the first).  One sweep's time instead of two.  Over 24 graphs (5k / 10k / 20k poses, eight seeds each,
  // tools/nd_root_eval.py): 451 launched tree levels in all against 460 with the double sweep (K = 2 / 3 / 5 / 6 / 8 / 16: 447 / 450 /
  // 449 / 464 / 460 / 456), 15 graphs the same height, 5 shorter, 4 taller (single graphs -9 .. +5), factorisation flops 1.000 x
  // on average -- the height of a dissection tree moves by chance with the root's cut, no start rule predicts it (DESIGN.md
  // 2.1); the benchmark graph happens to come out a level shorter (17 / 15 launched instead of 18 / 16).  The result does not
  // depend on the number of threads (the same four candidates whatever runs them).  CGMR_ND_ROOT_STARTS=0: the double sweep.
  static const int root_starts = getenv("CGMR_ND_ROOT_STARTS") ? atoi(getenv("CGMR_ND_ROOT_STARTS")) : 4;
  bool multi_done = false;
  int reached = 0;
  if (depth == 0 && !have_start && root_starts > 1 && n >= 2048) {
    const int K = std::min(root_starts, 16);
    struct Cand { std::vector<NDCtx::VState> vs; std::vector<int32_t> q, lv; int reached = 0, nlev = 0; };
    static thread_local std::vector<Cand> cand;
    cand.resize(K);
    Cand* const cands = cand.data();                     // (the helpers must not name the thread-local itself: theirs is another)
    std::vector<HelperPool::Job> jobs(K);
    auto run = [&, cands](int k) {
      Cand& X = cands[k];
      X.vs.assign(C.vs.begin(), C.vs.end());
      X.q.resize(n); X.lv.assign(n + 1, 0);
      const int s0 = C.order[begin + (int)((int64_t)k * (n - 1) / (K - 1))];
      X.reached = bfs(C, X.q.data(), s0, id, vis2, X.lv.data(), X.vs.data());
      X.nlev = X.vs[X.q[X.reached - 1]].dist + 1;
    };
    for (int k = 1; k < K; k++) { jobs[k].fn = [&run, k] { run(k); }; pool().run(jobs[k]); }
    run(0);
    for (int k = 1; k < K; k++) HelperPool::wait(jobs[k]);
    if (cand[0].reached == n) {                          // (a disconnected range takes the ordinary path below)
      int best = 0;
      for (int k = 1; k < K; k++) if (cand[k].nlev > cand[best].nlev) best = k;
      const Cand& X = cand[best];
      for (int q = 0; q < n; q++) { const int v = X.q[q]; C.vs[v] = X.vs[v]; Q[q] = v; }
      std::copy(X.lv.begin(), X.lv.begin() + X.nlev + 1, LV);
      reached = n;
      multi_done = true;
    }
  }
  if (!multi_done) reached = have_start ? bfs(C, Q, start, id, vis2, LV) : bfs(C, Q, C.order[begin], id, vis1);
  const double t_l1 = nd_trace ? now_s() : 0;
  if (reached < n) {
    // disconnected: component first, then the rest (independent subtrees, no separator)
    int k = begin;
    for (int q = 0; q < reached; q++) C.tmp[k++] = Q[q];
    for (int p = begin; p < end; p++) if (C.vs[C.order[p]].label == id) C.tmp[k++] = C.order[p];
    std::copy(C.tmp.begin() + begin, C.tmp.begin() + end, C.order.begin() + begin);
    int n1 = -1, n2 = -1;
    NDRange r1 = nd(C, begin, begin + reached, depth, -1, &n1);   // (passing the parent's end vertex on instead of a fresh double sweep: 466 instead of 448 tree levels over 24 graphs -- round 6)
    NDRange r2 = nd(C, begin + reached, end, depth, -1, &n2);
    r2.height = std::max(r1.height, r2.height);
    node = C.new_rec(n1, n2, n, 0);                       // independent components: two halves, no separator
    return r2;
  }
  if (!have_start && !multi_done) {
    int far = Q[reached - 1];
    // second sweep from the far vertex gives the level structure
    bfs(C, Q, far, vis1, vis2, LV);
  }
  const double t_l2 = nd_trace ? now_s() : 0;
  int nlev = C.vs[Q[n - 1]].dist + 1;
  if (nlev <= 2) { node = C.new_rec(-1, -1, n, n); return emit_panels(C, begin, end); }  // clique-like: nothing to dissect
  thread_local std::vector<int32_t> lvl_cnt;            // (the sweep counted the levels into the partition scratch: keep a copy)
  lvl_cnt.assign(LV, LV + nlev);
  lvl_cnt.push_back(0);
  // choose the separator level
  int best = -1, best_sz = 1 << 30, fallback = 1, fb_bal = -1;
  int cum = lvl_cnt[0];
  for (int j = 1; j <= nlev - 2; j++) {
    int a = cum, b = n - cum - lvl_cnt[j];
    int bal = std::min(a, b);
    if (bal > fb_bal) { fb_bal = bal; fallback = j; }
    if (bal * 10 >= n * 3 && lvl_cnt[j] < best_sz) { best_sz = lvl_cnt[j]; best = j; }
    cum += lvl_cnt[j];
  }
  int js = best >= 0 ? best : fallback;
  // Separator: not all of level js, but a minimum vertex cover of the edges between level js and level js+1 (Koenig:
  // from a maximum bipartite matching) -- every path from the near side to the far side crosses one of those edges.
  // X = level-js vertices with a neighbour in level js+1, Y = those neighbours.  Covered vertices get dist = js,
  // the other X move to the near side (dist js-1), the other Y stay on the far side.
  {
    int s_js = 0;
    for (int j = 0; j < js; j++) s_js += lvl_cnt[j];
    thread_local std::vector<int32_t> ylocal;             // vertex -> index in Y, -1 outside this block
    if ((int)ylocal.size() < (int)C.vs.size()) ylocal.assign(C.vs.size(), -1);
    std::vector<int32_t> X, Y, xptr(1, 0), xadj;
    for (int q = s_js; q < s_js + lvl_cnt[js]; q++) {
      int v = Q[q];
      const size_t before = xadj.size();
      for (int p = C.ap[v]; p < C.ap[v + 1]; p++) {
        int w = C.ai[p];
        if (C.vs[w].label != vis2 || C.vs[w].dist != js + 1) continue;
        if (ylocal[w] < 0) { ylocal[w] = (int)Y.size(); Y.push_back(w); }
        xadj.push_back(ylocal[w]);
      }
      if (xadj.size() == before) { C.vs[v].dist = js - 1; continue; }   // touches nothing beyond: near side
      X.push_back(v);
      xptr.push_back((int)xadj.size());
    }
    static const bool min_cover = !(getenv("CGMR_ND_MIN_COVER") && atoi(getenv("CGMR_ND_MIN_COVER")) == 0);
    const int nx = (int)X.size(), ny = (int)Y.size();
    if (min_cover && nx > 1 && ny > 0) {
      std::vector<int32_t> mx(nx, -1), my(ny, -1), seen(ny, -1), stack;
      // maximum matching by augmenting paths (the blocks are small: tens of vertices)
      std::function<bool(int, int)> augment = [&](int x, int stamp) {
        for (int p = xptr[x]; p < xptr[x + 1]; p++) {
          int y = xadj[p];
          if (seen[y] == stamp) continue;
          seen[y] = stamp;
          if (my[y] < 0 || augment(my[y], stamp)) { mx[x] = y; my[y] = x; return true; }
        }
        return false;
      };
      for (int x = 0; x < nx; x++) augment(x, x);
      // Koenig: Z = reachable from the unmatched X by alternating paths; cover = (X \ Z) + (Y in Z)
      std::vector<uint8_t> zx(nx, 0), zy(ny, 0);
      for (int x = 0; x < nx; x++) if (mx[x] < 0) { zx[x] = 1; stack.push_back(x); }
      while (!stack.empty()) {
        int x = stack.back(); stack.pop_back();
        for (int p = xptr[x]; p < xptr[x + 1]; p++) {
          int y = xadj[p];
          if (zy[y] || mx[x] == y) continue;
          zy[y] = 1;
          int x2 = my[y];
          if (x2 >= 0 && !zx[x2]) { zx[x2] = 1; stack.push_back(x2); }
        }
      }
      for (int x = 0; x < nx; x++) if (zx[x]) C.vs[X[x]].dist = js - 1;      // not in the cover: near side
      for (int y = 0; y < ny; y++) if (zy[y]) C.vs[Y[y]].dist = js;          // in the cover: separator
    }
    for (int w : Y) ylocal[w] = -1;
  }
  // The queue is sorted by level, and the cover only moved vertices of level js (to js-1) and of level js+1 (to js):
  // everything before the level-js block is near side, everything behind the level-(js+1) block far side; only the
  // two blocks are looked at vertex by vertex.  Same order inside the three parts as a pass over the whole queue.
  const double t_l3 = nd_trace ? now_s() : 0;
  int s_js = 0;
  for (int j = 0; j < js; j++) s_js += lvl_cnt[j];
  const int e_js = s_js + lvl_cnt[js], e_js1 = e_js + lvl_cnt[js + 1];
  int na = s_js, nb = n - e_js1;
  for (int q = s_js; q < e_js; q++) if (C.vs[Q[q]].dist < js) na++;
  for (int q = e_js; q < e_js1; q++) if (C.vs[Q[q]].dist > js) nb++;
  // the two ends of this range, where the halves' sweeps start: the root of the sweep and the farthest vertex that
  // stayed on the far side (the cover may have claimed the last ones for the separator)
  const int start_a = Q[0];
  int start_b = -1;
  for (int q = n - 1; q >= 0 && start_b < 0; q--) if (C.vs[Q[q]].dist > js) start_b = Q[q];
  int pa = begin, pb = begin + na, ps = begin + na + nb;
  for (int q = 0; q < s_js; q++) C.tmp[pa++] = Q[q];
  for (int q = s_js; q < e_js; q++) { const int v = Q[q]; if (C.vs[v].dist < js) C.tmp[pa++] = v; else C.tmp[ps++] = v; }
  for (int q = e_js; q < e_js1; q++) { const int v = Q[q]; if (C.vs[v].dist > js) C.tmp[pb++] = v; else C.tmp[ps++] = v; }
  for (int q = e_js1; q < n; q++) C.tmp[pb++] = Q[q];
  std::copy(C.tmp.begin() + begin, C.tmp.begin() + end, C.order.begin() + begin);
  if (nd_trace && depth <= 3) fprintf(stderr, "    nd depth %d n %5d own work %.1f us (labels %.1f, sweep 1 %.1f, sweep 2 %.1f, level choice + cover %.1f, partition %.1f; levels %d, separator level %d of size %d)\n", depth, n, 1e6 * (now_s() - t_in), 1e6 * (t_l0 - t_in), 1e6 * (t_l1 - t_l0), 1e6 * (t_l2 - t_l1), 1e6 * (t_l3 - t_l2), 1e6 * (now_s() - t_l3), nlev, js, lvl_cnt[js]);
  // the two halves touch disjoint vertices and disjoint position ranges: fork one of them near the top
  NDRange r1, r2;
  int n1 = -1, n2 = -1;
  if (depth < C.max_par_depth && na > 512 && nb > 512) {
    HelperPool::Job job;
    job.fn = [&C, &r1, &n1, begin, na, depth, start_a] { r1 = nd(C, begin, begin + na, depth + 1, start_a, &n1); };
    pool().run(job);
    r2 = nd(C, begin + na, begin + na + nb, depth + 1, start_b, &n2);
    HelperPool::wait(job);
  } else {
    r1 = nd(C, begin, begin + na, depth + 1, start_a, &n1);
    r2 = nd(C, begin + na, begin + na + nb, depth + 1, start_b, &n2);
  }
  // Only the half right in front of the separator can share a panel with it (see the amalgamation in analyze()):
  // that should be the taller one, so the two blocks trade places when the first turned out taller.
  const int s0 = begin + na + nb;
  const bool swapped = r1.height > r2.height;
  if (swapped) {
    std::rotate(C.order.begin() + begin, C.order.begin() + begin + na, C.order.begin() + s0);
    std::rotate(C.pstart.begin() + begin, C.pstart.begin() + begin + na, C.pstart.begin() + s0);
    if (depth < kRangeDepth) {                           // subtree ranges recorded below move with their blocks
      std::lock_guard<std::mutex> lk(C.range_mu);
      for (auto& pr : C.subtree_ranges) {
        if (pr.first >= begin && pr.second <= begin + na) { pr.first += nb; pr.second += nb; }
        else if (pr.first >= begin + na && pr.second <= s0) { pr.first -= na; pr.second -= na; }
      }
    }
    std::swap(r1, r2);
    std::swap(n1, n2);
  }
  node = C.new_rec(n1, n2, n, end - s0);
  if (depth == kRangeDepth && na > 0 && nb > 0) {
    const int mid = begin + (swapped ? nb : na);
    std::lock_guard<std::mutex> lk(C.range_mu);
    C.subtree_ranges.emplace_back(begin, mid);
    C.subtree_ranges.emplace_back(mid, s0);
  }
  // the separator in runs of kPanelW; when its remainder fits into the last panel of the half in front of it, the
  // remainder goes first so that the amalgamation can merge the two (one level less on that path)
  const int run = kPanelW;
  const int ssz = end - s0, rem = ssz % run;
  NDRange out;
  if (rem != 0 && r2.last + rem <= run) {
    C.pstart[s0] = 1;
    NDRange rest = emit_panels(C, s0 + rem, end, run);
    out.last = ssz == rem ? r2.last + rem : rest.last;
    out.height = std::max(r1.height + 1, r2.height) + rest.height;
  } else {
    NDRange all = emit_panels(C, s0, end, run);
    out.last = all.last;
    out.height = std::max(r1.height, r2.height) + all.height;
  }
  if (nd_trace && depth <= 4) fprintf(stderr, "    nd depth %d n %5d subtree done after %.1f us (halves %d + %d, forked %d)\n", depth, n, 1e6 * (now_s() - t_in), na, nb, (depth < C.max_par_depth && na > 512 && nb > 512) ? 1 : 0);
  return out;
}

// Adjacency rows are short (a pose has ~8 neighbours).  Up to 16 entries go through a sorting network on a padded copy
// (Batcher's merge exchange: 19 compare-exchanges for 8 values, 63 for 16; minimum / maximum pairs, no branch that depends on
// the data -- the insertion sort this replaces mispredicted its inner loop's exit about once per entry: 12 ns per entry of the
// permuted adjacency), longer rows through std::sort.
#define CGMR_CE(a, b) do { const int32_t lo_ = v[a] < v[b] ? v[a] : v[b], hi_ = v[a] < v[b] ? v[b] : v[a]; v[a] = lo_; v[b] = hi_; } while (0)
inline void sort_net8(int32_t* v) {
  CGMR_CE(0,4); CGMR_CE(1,5); CGMR_CE(2,6); CGMR_CE(3,7); CGMR_CE(0,2); CGMR_CE(1,3); CGMR_CE(4,6); CGMR_CE(5,7); CGMR_CE(2,4); CGMR_CE(3,5);
  CGMR_CE(0,1); CGMR_CE(2,3); CGMR_CE(4,5); CGMR_CE(6,7); CGMR_CE(1,4); CGMR_CE(3,6); CGMR_CE(1,2); CGMR_CE(3,4); CGMR_CE(5,6);
}
inline void sort_net16(int32_t* v) {
  CGMR_CE(0,8); CGMR_CE(1,9); CGMR_CE(2,10); CGMR_CE(3,11); CGMR_CE(4,12); CGMR_CE(5,13); CGMR_CE(6,14); CGMR_CE(7,15);
  CGMR_CE(0,4); CGMR_CE(1,5); CGMR_CE(2,6); CGMR_CE(3,7); CGMR_CE(8,12); CGMR_CE(9,13); CGMR_CE(10,14); CGMR_CE(11,15);
  CGMR_CE(4,8); CGMR_CE(5,9); CGMR_CE(6,10); CGMR_CE(7,11); CGMR_CE(0,2); CGMR_CE(1,3); CGMR_CE(4,6); CGMR_CE(5,7);
  CGMR_CE(8,10); CGMR_CE(9,11); CGMR_CE(12,14); CGMR_CE(13,15); CGMR_CE(2,8); CGMR_CE(3,9); CGMR_CE(6,12); CGMR_CE(7,13);
  CGMR_CE(2,4); CGMR_CE(3,5); CGMR_CE(6,8); CGMR_CE(7,9); CGMR_CE(10,12); CGMR_CE(11,13); CGMR_CE(0,1); CGMR_CE(2,3);
  CGMR_CE(4,5); CGMR_CE(6,7); CGMR_CE(8,9); CGMR_CE(10,11); CGMR_CE(12,13); CGMR_CE(14,15); CGMR_CE(1,8); CGMR_CE(3,10);
  CGMR_CE(5,12); CGMR_CE(7,14); CGMR_CE(1,4); CGMR_CE(3,6); CGMR_CE(5,8); CGMR_CE(7,10); CGMR_CE(9,12); CGMR_CE(11,14);
  CGMR_CE(1,2); CGMR_CE(3,4); CGMR_CE(5,6); CGMR_CE(7,8); CGMR_CE(9,10); CGMR_CE(11,12); CGMR_CE(13,14);
}
#undef CGMR_CE
inline void sort_row(int32_t* b, int32_t* e) {
  const int n = (int)(e - b);
  if (n <= 1) return;
  if (n > 16) { std::sort(b, e); return; }
  int32_t v[16];
  if (n <= 8) {
    for (int u = 0; u < 8; u++) v[u] = 0x7fffffff;
    for (int u = 0; u < n; u++) v[u] = b[u];
    sort_net8(v);
  } else {
    for (int u = 0; u < 16; u++) v[u] = 0x7fffffff;
    for (int u = 0; u < n; u++) v[u] = b[u];
    sort_net16(v);
  }
  for (int u = 0; u < n; u++) b[u] = v[u];
}

// run fn(lo, hi) over [0, n) on up to nthreads threads (static split)
template <typename Fn>
void parallel_for(int n, int nthreads, Fn&& fn, int min_n = 4096) {
  if (nthreads <= 1 || n < min_n) { fn(0, n); return; }
  std::vector<HelperPool::Job> jobs(nthreads - 1);
  int per = (n + nthreads - 1) / nthreads;
  for (int t = 1; t < nthreads; t++) {
    int lo = t * per, hi = std::min(n, lo + per);
    jobs[t - 1].fn = [&fn, lo, hi] { if (lo < hi) fn(lo, hi); };
    pool().run(jobs[t - 1]);
  }
  fn(0, std::min(n, per));
  for (auto& j : jobs) HelperPool::wait(j);
}

}  // namespace

// run task(0) .. task(n - 1) on the helper pool (the caller takes the last one); for the callers of analyze() that have
// a few independent host-side jobs of their own
void host_run_tasks(int n, const std::function<void(int)>& task) {
  if (n <= 0) return;
  std::vector<HelperPool::Job> jobs(n - 1);
  for (int t = 0; t + 1 < n; t++) {
    jobs[t].fn = [&task, t] { task(t); };
    pool().run(jobs[t]);
  }
  task(n - 1);
  for (auto& j : jobs) HelperPool::wait(j);
}

namespace {

// CPUs' worth of time the process may use per scheduling period when its control group caps it (cgroup v2 cpu.max, v1
// cpu.cfs_quota_us / cpu.cfs_period_us); 0 = no cap found
double cgroup_cpu_quota() {
  double q = 0;
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char a[64] = {0};
    long long period = 0;
    if (fscanf(f, "%63s %lld", a, &period) == 2 && strcmp(a, "max") != 0 && period > 0) q = (double)atoll(a) / (double)period;
    fclose(f);
    return q;
  }
  long long quota = -1, period = 0;
  if (FILE* f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(f, "%lld", &quota) != 1) quota = -1; fclose(f); }
  if (FILE* f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(f, "%lld", &period) != 1) period = 0; fclose(f); }
  return (quota > 0 && period > 0) ? (double)quota / (double)period : 0;
}

int host_threads() {
  static int n = [] {
    const char* e = getenv("CGMR_HOST_THREADS");
    int v = e ? atoi(e) : 0;
    if (v <= 0) {
      unsigned hc = std::thread::hardware_concurrency();
      v = hc >= 32 ? 8 : (hc >= 8 ? 4 : (hc >= 4 ? 2 : 1));
      // A container sees all the host's CPUs and may still be capped to a few CPUs' worth of time (the GPU boxes: 256
      // hardware threads, cpu.max = 16 CPUs): threads beyond the cap are throttled, not run -- eight ranks with eight
      // spinning helpers each took 3.8 s per C5 round there (round 4).  The ranks of one node share the cap.
      const double quota = cgroup_cpu_quota();
      if (quota > 0) {
        const char* lw = getenv("LOCAL_WORLD_SIZE") ? getenv("LOCAL_WORLD_SIZE") : getenv("WORLD_SIZE");
        const int ranks = std::max(1, lw ? atoi(lw) : 1);
        v = std::min(v, std::max(1, (int)(quota / ranks)));
      }
    }
    return std::max(1, std::min(v, 16));
  }();
  return n;
}

}  // namespace

void host_pool_info(int out[5]) {
  HelperPool& p = pool();
  cpu_set_t allowed;
  CPU_ZERO(&allowed);
  out[0] = p.helpers() + 1;
  out[1] = p.home_cpu() >= 0 ? 1 : 0;
  out[2] = p.home_cpu();
  out[3] = sched_getaffinity(0, sizeof allowed, &allowed) == 0 ? CPU_COUNT(&allowed) : -1;
  out[4] = p.moves();
}

namespace {

// the node table of a from-scratch ordering, from what analyze() kept of it (see Symbolic::nd_pending)
void materialize_nd_tree(Symbolic& S) {
  if (!S.nd_pending) return;
  S.nd_pending = false;
  const std::vector<Symbolic::NDRecLite>& recs = S.nd_pending_recs;
  const std::vector<int32_t>& order = S.nd_pending_order;
  S.nd_nodes.clear();
  S.nd_nodes.reserve(recs.size() + 8);
  std::function<int(int, int, int)> build = [&](int rec, int begin, int parent) -> int {
    if (rec < 0 || rec >= (int)recs.size()) return -1;
    const Symbolic::NDRecLite R = recs[rec];
    const int id = (int)S.nd_nodes.size();
    S.nd_nodes.emplace_back();
    S.nd_nodes[id].parent = parent;
    S.nd_nodes[id].count = R.n;
    const int na = R.a >= 0 ? recs[R.a].n : 0, nb = R.b >= 0 ? recs[R.b].n : 0;
    const int ca = build(R.a, begin, id), cb = build(R.b, begin + na, id);
    S.nd_nodes[id].a = ca; S.nd_nodes[id].b = cb;
    S.nd_nodes[id].verts.assign(order.begin() + begin + na + nb, order.begin() + begin + na + nb + R.ssz);
    return id;
  };
  S.nd_root = build(S.nd_pending_root, 0, -1);
  if (!S.nd_pending_hubs.empty()) {
    if (S.nd_root < 0) { S.nd_nodes.emplace_back(); S.nd_root = (int)S.nd_nodes.size() - 1; S.nd_nodes[S.nd_root].parent = -1; }
    Symbolic::NDNode& Rn = S.nd_nodes[S.nd_root];
    Rn.verts.insert(Rn.verts.end(), S.nd_pending_hubs.begin(), S.nd_pending_hubs.end());
    Rn.count += (int)S.nd_pending_hubs.size();
  }
  std::vector<Symbolic::NDRecLite>().swap(S.nd_pending_recs);
  std::vector<int32_t>().swap(S.nd_pending_order);
  std::vector<int32_t>().swap(S.nd_pending_hubs);
}

// ------------------------------------------------------------------ incremental ordering
// The previous dissection tree extended by the vertices prev.nf .. nf-1 (gn_symbolic.h: Symbolic::NDNode).  A vertex may live
// in node X (a leaf, or the separator of an inner node) iff each of its neighbours lives in X's subtree or in the
// separator of an ancestor of X -- then everything that couples two subtrees still sits in a separator above both.  The
// new vertices are placed in index order (a new pose's neighbours are older poses and poses placed just before it):
//   neighbours' nodes minus those that are ancestors of another one = the constraint set
//   one node, a leaf            -> into that leaf
//   one node, an inner one      -> below it: down the lighter child to a leaf (the neighbour sits in the separator above)
//   several nodes               -> into the separator of their lowest common ancestor
// A leaf that outgrows two panels is dissected again on its own vertices (a robot that keeps exploring appends to the same
// leaf: without this the tree would grow a chain of panels there, one level per 16 poses).  Then the order and the panel
// starts are emitted by a walk over the tree that cuts panels exactly like nd() does.  Returns false if the tree cannot
// take the vertices (order / pstart / pos_ranges are not modified then; prev's tree is consumed either way).
bool extend_order(Symbolic& prev, int nf, const std::vector<int32_t>& ap, const std::vector<int32_t>& ai,
                  std::vector<int32_t>& order, std::vector<uint8_t>& pstart, std::vector<std::pair<int, int>>& pos_ranges,
                  Symbolic& S, int n_new_edges, const int32_t* new_ef, const int32_t* new_et, const std::vector<int32_t>& hidx,
                  const std::vector<int32_t>& forced_hubs) {
  typedef Symbolic::NDNode Node;
  materialize_nd_tree(prev);
  std::vector<Node> T = std::move(prev.nd_nodes);            // (the caller's previous analysis is discarded afterwards either way)
  const int root = prev.nd_root;
  if (root < 0 || T.empty()) return false;
  std::vector<int32_t> where(nf, -1), depth(T.size(), 0);
  {
    std::vector<int> stack(1, root);
    while (!stack.empty()) {
      const int x = stack.back(); stack.pop_back();
      for (int32_t v : T[x].verts) { if (v < 0 || v >= nf) return false; where[v] = x; }
      for (int c : {T[x].a, T[x].b}) if (c >= 0) { depth[c] = depth[x] + 1; stack.push_back(c); }
    }
  }
  auto is_ancestor = [&](int anc, int x) {               // anc == x counts
    while (x >= 0 && depth[x] > depth[anc]) x = T[x].parent;
    return x == anc;
  };
  auto lca = [&](int x, int y) {
    while (x != y) { if (depth[x] >= depth[y]) x = T[x].parent; else y = T[y].parent; }
    return x;
  };
  // ---- forced hubs (the caller names them: the gauge vertices of the condensed stars received from the peers) live in the
  // root's separator: a vertex may always move UP the tree (it is eliminated later; whatever it couples is then coupled
  // through a separator above both), and an edge to a vertex of the root lies on a root path whatever its other end
  std::vector<uint8_t> placed(nf, 0);
  for (int h : forced_hubs) {
    if (h < 0 || h >= nf) continue;
    const int x = where[h];
    if (x == root) continue;
    if (x >= 0) {
      std::vector<int32_t>& vx = T[x].verts;
      vx.erase(std::find(vx.begin(), vx.end(), h));
      for (int y = x; y >= 0 && y != root; y = T[y].parent) T[y].count--;
    } else {
      T[root].count++;
    }
    T[root].verts.push_back(h);
    where[h] = root;
    placed[h] = 1;
  }
  std::vector<int> grown;                                 // leaves to look at again
  std::vector<int> cand;
  for (int v = prev.nf; v < nf; v++) {
    if (placed[v]) continue;
    cand.clear();
    for (int p = ap[v]; p < ap[v + 1]; p++) {
      const int x = where[ai[p]];
      if (x >= 0 && std::find(cand.begin(), cand.end(), x) == cand.end()) cand.push_back(x);
    }
    int target;
    if (cand.empty()) target = root;                       // no placed neighbour: anywhere; the root's separator couples nothing
    else {
      // drop every node that is an ancestor of another candidate
      std::vector<int> low;
      for (int x : cand) {
        bool anc = false;
        for (int y : cand) if (y != x && is_ancestor(x, y)) { anc = true; break; }
        if (!anc) low.push_back(x);
      }
      target = low[0];
      for (size_t k = 1; k < low.size(); k++) target = lca(target, low[k]);
      if (low.size() == 1) {
        while (T[target].a >= 0 || T[target].b >= 0) {     // below an inner node: the lighter child, down to a leaf
          const int a = T[target].a, b = T[target].b;
          target = (a >= 0 && (b < 0 || T[a].count <= T[b].count)) ? a : b;
        }
      }
    }
    T[target].verts.push_back(v);
    where[v] = target;
    for (int x = target; x >= 0; x = T[x].parent) T[x].count++;
    if (T[target].a < 0 && T[target].b < 0 && (int)T[target].verts.size() > 2 * kPanelW &&
        std::find(grown.begin(), grown.end(), target) == grown.end()) grown.push_back(target);
  }
  // ---- the new edges may also join two vertices the tree already held (a loop closure between old poses, a condensed edge
  // from a peer): their nodes must lie on one root path, or the two subtrees are not independent any more -- the parallel
  // border computation below hands "independent" subtrees to different threads and produced a wrong structure for such a
  // graph (found with the C5 rounds of two robots with a context each: Cholesky failure in round 40).  Then: full analysis.
  static const bool check_new_edges = !(getenv("CGMR_SYM_EXTEND_CHECK") && atoi(getenv("CGMR_SYM_EXTEND_CHECK")) == 0);   // (0: the round-3 bug, for its test)
  for (int k = 0; check_new_edges && k < n_new_edges; k++) {
    const int a = hidx[new_ef[k]], b = hidx[new_et[k]];
    if (a < 0 || b < 0 || a == b) continue;
    const int x = where[a], y = where[b];
    if (x < 0 || y < 0) return false;
    if (x != y && !is_ancestor(x, y) && !is_ancestor(y, x)) return false;
  }
  // ---- leaves that outgrew two panels: dissect them again (their own vertices, the full adjacency)
  if (!grown.empty()) {
    std::vector<int32_t> lorder;
    std::vector<uint8_t> lpstart;
    for (int leaf : grown) {
      const int n = (int)T[leaf].verts.size();
      lorder = T[leaf].verts;
      lpstart.assign(n, 0);
      NDCtx C{ap, ai, lorder, lpstart};
      C.vs.assign(nf, NDCtx::VState{0, 0});
      C.queue.assign(n, 0);
      C.tmp.assign(n, 0);
      C.recs.assign(2 * (size_t)n + 4, NDCtx::Rec());
      C.max_par_depth = 0;
      int root_rec = -1;
      nd(C, 0, n, kRangeDepth + 1, -1, &root_rec);         // (depth beyond kRangeDepth: records no subtree ranges)
      if (root_rec < 0) continue;
      const int parent = T[leaf].parent;
      std::function<int(int, int, int, int)> build = [&](int rec, int begin, int par, int reuse) -> int {
        if (rec < 0) return -1;
        const NDCtx::Rec R = C.recs[rec];
        int id = reuse;
        if (id < 0) { id = (int)T.size(); T.emplace_back(); depth.push_back(0); }
        T[id].parent = par;
        T[id].count = R.n;
        depth[id] = par >= 0 ? depth[par] + 1 : 0;
        const int na = R.a >= 0 ? C.recs[R.a].n : 0, nb = R.b >= 0 ? C.recs[R.b].n : 0;
        const int ca = build(R.a, begin, id, -1), cb = build(R.b, begin + na, id, -1);
        T[id].a = ca; T[id].b = cb;
        T[id].verts.assign(lorder.begin() + begin + na + nb, lorder.begin() + begin + na + nb + R.ssz);
        return id;
      };
      build(root_rec, 0, parent, leaf);
    }
  }
  // ---- order + panel starts: children, then the separator, panels cut as in nd()
  struct Emit {
    std::vector<Node>& T;
    std::vector<int32_t>& order;
    std::vector<uint8_t>& pstart;
    std::vector<std::pair<int, int>>& ranges;
    int pos = 0;
    NDRange panels(int begin, int end) {
      NDRange r;
      for (int p = begin; p < end; p += kPanelW) { pstart[p] = 1; r.height++; }
      r.last = end > begin ? (end - begin - 1) % kPanelW + 1 : 0;
      return r;
    }
    NDRange walk(int x, int d) {
      if (x < 0) return NDRange();
      const Node& X = T[x];
      const int begin = pos;
      if (X.a < 0 && X.b < 0) {
        for (int32_t v : X.verts) order[pos++] = v;
        return panels(begin, pos);
      }
      const NDRange r1 = walk(X.a, d + 1);
      const int mid = pos;
      const NDRange r2 = walk(X.b, d + 1);
      const int s0 = pos;
      if (d == kRangeDepth && mid > begin && s0 > mid) { ranges.emplace_back(begin, mid); ranges.emplace_back(mid, s0); }
      for (int32_t v : X.verts) order[pos++] = v;
      const int end = pos, ssz = end - s0, rem = ssz % kPanelW;
      NDRange out;
      if (ssz == 0) { out = r2; out.height = std::max(r1.height, r2.height); return out; }
      if (rem != 0 && X.b >= 0 && r2.last + rem <= kPanelW) {
        pstart[s0] = 1;
        const NDRange rest = panels(s0 + rem, end);
        out.last = ssz == rem ? r2.last + rem : rest.last;
        out.height = std::max(r1.height + 1, r2.height) + rest.height;
      } else {
        const NDRange all = panels(s0, end);
        out.last = all.last;
        out.height = std::max(r1.height, r2.height) + all.height;
      }
      return out;
    }
  };
  // (into temporaries: `order` and `pstart` stay the caller's identity order / zeros unless the tree takes every vertex
  // exactly once -- the from-scratch ordering that follows a refusal starts from them)
  {
    size_t held = 0;
    std::vector<int> stack(1, root);
    while (!stack.empty()) {
      const int x = stack.back(); stack.pop_back();
      held += T[x].verts.size();
      for (int c : {T[x].a, T[x].b}) if (c >= 0) stack.push_back(c);
    }
    if (held != (size_t)nf) return false;
  }
  std::vector<int32_t> order2(nf);
  std::vector<uint8_t> pstart2(nf, 0);
  std::vector<std::pair<int, int>> ranges2;
  Emit E{T, order2, pstart2, ranges2};
  const NDRange whole = E.walk(root, 0);
  if (E.pos != nf) return false;
  // The device pays per tree level (one factor + one update launch and a backward hop, ~26 us per Gauss-Newton pass), an
  // extension saves ~0.35 ms of ordering: a tree that has grown more than a few panels taller than the last from-scratch
  // ordering's is not worth keeping (CGMR_SYM_EXTEND_SLACK panels, default 2; the simulated C5 rounds of eight robots: mean
  // tree height 13.5 levels extended without the bound, 11.2 from scratch every round).
  static const int slack = getenv("CGMR_SYM_EXTEND_SLACK") ? atoi(getenv("CGMR_SYM_EXTEND_SLACK")) : 2;
  if (slack >= 0 && prev.nd_height_full > 0 && whole.height > prev.nd_height_full + slack) return false;
  S.nd_height_full = prev.nd_height_full;
  order.swap(order2);
  pstart.swap(pstart2);
  pos_ranges.swap(ranges2);
  S.nd_nodes.swap(T);
  S.nd_root = root;
  S.nd_nf_full = prev.nd_nf_full;
  S.nd_appended = prev.nd_appended + (nf - prev.nf);
  return true;
}

}  // namespace

int analyze(int nV, const uint8_t* fixed, int nE, const int32_t* ef, const int32_t* et, Symbolic& S, Symbolic* prev,
            int n_common, const int32_t* hub_vertices, int n_hub_vertices, const AnalyzeHooks* hooks) {
  double t0 = now_s();
  static const bool trace = getenv("CGMR_SYM_TRACE") != nullptr;
  struct Rebalance {                                             // (destroyed after at_home: the caller's own mask is back)
    long long w0 = HelperPool::now_ns(CLOCK_MONOTONIC), c0 = HelperPool::now_ns(CLOCK_THREAD_CPUTIME_ID);
    ~Rebalance() {
      pool().account_caller(HelperPool::now_ns(CLOCK_MONOTONIC) - w0, HelperPool::now_ns(CLOCK_THREAD_CPUTIME_ID) - c0);
      pool().rebalance();
    }
  } rebalance_when_done;
  HelperPool::CallerAtHome at_home(pool());
  double tc = t0;
  auto CK = [&](const char* what) { if (trace) { double t = now_s(); fprintf(stderr, "  sym %-28s %7.1f us\n", what, 1e6 * (t - tc)); tc = t; } };
  S = Symbolic();
  CK("  adj: previous analysis released");
  S.nV = nV;
  S.nE = nE;
  // the edge list checked and the vertices with an edge marked, by slices of the list (every mark is the same byte value: the
  // stores of two threads to one vertex do not care about their order)
  std::vector<uint8_t> active(nV, 0);
  {
    std::atomic<int> bad{0};
    uint8_t* act = active.data();
    parallel_for(nE, host_threads(), [&](int lo, int hi) {
      for (int k = lo; k < hi; k++) {
        if (ef[k] < 0 || ef[k] >= nV || et[k] < 0 || et[k] >= nV) { bad.store(1, std::memory_order_relaxed); continue; }
        __atomic_store_n(act + ef[k], (uint8_t)1, __ATOMIC_RELAXED);
        __atomic_store_n(act + et[k], (uint8_t)1, __ATOMIC_RELAXED);
      }
    }, 16384);
    if (bad.load()) return -1;
  }
  S.hidx.assign(nV, -1);
  int nf = 0;
  for (int v = 0; v < nV; v++) if (active[v] && !(fixed && fixed[v])) S.hidx[v] = nf++;
  S.nf = nf;
  S.vperm.assign(nV, -1);
  if (nf == 0) { S.level_ptr.assign(1, 0); return 0; }
  CK("  adj: checks, active, block indices");
  // adjacency CSR over block indices (deduplicated, sorted)
  // (counting and filling: every thread walks the whole edge list and takes the end points that fall into its own range of
  // block indices -- the reads are shared and sequential, the scattered writes are private: no atomics, same lists as a
  // one-thread pass)
  std::vector<int32_t> ap(nf + 1, 0), ai;
  const int NTA = (nE >= 16384 && nf >= 2048) ? host_threads() : 1;
  auto in_ranges = [&](auto&& body) {
    if (NTA <= 1) { body(0, nf); return; }
    std::vector<HelperPool::Job> jobs(NTA - 1);
    const int per = (nf + NTA - 1) / NTA;
    for (int t = 1; t < NTA; t++) {
      const int lo = t * per, hi = std::min(nf, lo + per);
      jobs[t - 1].fn = [&body, lo, hi] { if (lo < hi) body(lo, hi); };
      pool().run(jobs[t - 1]);
    }
    body(0, std::min(nf, per));
    for (auto& j : jobs) HelperPool::wait(j);
  };
  // Edge slices (round 5): every thread counts the end points of ITS share of the edge list into a column of its own, a pass
  // over the vertices turns the columns into row starts and per-thread write positions, every thread files its share -- the
  // rows hold their entries in edge order, exactly as the one-thread pass leaves them, and the edge list is read once per
  // pass instead of once per thread and pass (CGMR_ADJ_SLICES=0: every thread walks the whole list and keeps its vertex range).
  static const bool adj_slices = !(getenv("CGMR_ADJ_SLICES") && atoi(getenv("CGMR_ADJ_SLICES")) == 0);
  if (adj_slices && NTA > 1) {
    std::vector<int32_t> cnt((size_t)NTA * nf, 0);
    auto in_slices = [&](auto&& body) {
      std::vector<HelperPool::Job> jobs(NTA - 1);
      for (int t = 1; t < NTA; t++) {
        jobs[t - 1].fn = [&body, t] { body(t); };
        pool().run(jobs[t - 1]);
      }
      body(0);
      for (auto& j : jobs) HelperPool::wait(j);
    };
    auto slice = [&](int t, int& lo, int& hi) { lo = (int)((int64_t)nE * t / NTA); hi = (int)((int64_t)nE * (t + 1) / NTA); };
    in_slices([&](int t) {
      int lo, hi;
      slice(t, lo, hi);
      int32_t* c = cnt.data() + (size_t)t * nf;
      for (int k = lo; k < hi; k++) {
        const int a = S.hidx[ef[k]], b = S.hidx[et[k]];
        if (a < 0 || b < 0 || a == b) continue;
        c[a]++; c[b]++;
      }
    });
    for (int v = 0; v < nf; v++) {                       // row starts; cnt[t][v] becomes thread t's first position in row v
      int32_t at = ap[v];
      for (int t = 0; t < NTA; t++) { const int32_t n = cnt[(size_t)t * nf + v]; cnt[(size_t)t * nf + v] = at; at += n; }
      ap[v + 1] = at;
    }
    ai.resize(ap[nf]);
    in_slices([&](int t) {
      int lo, hi;
      slice(t, lo, hi);
      int32_t* pos = cnt.data() + (size_t)t * nf;
      for (int k = lo; k < hi; k++) {
        const int a = S.hidx[ef[k]], b = S.hidx[et[k]];
        if (a < 0 || b < 0 || a == b) continue;
        ai[pos[a]++] = b;
        ai[pos[b]++] = a;
      }
    });
  } else {
  in_ranges([&](int lo, int hi) {
    for (int k = 0; k < nE; k++) {
      const int a = S.hidx[ef[k]], b = S.hidx[et[k]];
      if (a < 0 || b < 0 || a == b) continue;
      if (a >= lo && a < hi) ap[a + 1]++;
      if (b >= lo && b < hi) ap[b + 1]++;
    }
  });
  for (int v = 0; v < nf; v++) ap[v + 1] += ap[v];
  ai.resize(ap[nf]);
  {
    std::vector<int32_t> pos(ap.begin(), ap.end() - 1);
    in_ranges([&](int lo, int hi) {
      for (int k = 0; k < nE; k++) {
        const int a = S.hidx[ef[k]], b = S.hidx[et[k]];
        if (a < 0 || b < 0 || a == b) continue;
        if (a >= lo && a < hi) ai[pos[a]++] = b;
        if (b >= lo && b < hi) ai[pos[b]++] = a;
      }
    });
  }
  }
  CK("  adj: count + file");
  {
    // sort + dedupe each row (in parallel), then compact in place
    std::vector<int32_t> len(nf);
    parallel_for(nf, host_threads(), [&](int lo, int hi) {
      for (int v = lo; v < hi; v++) {
        int b = ap[v], e = ap[v + 1];
        sort_row(ai.data() + b, ai.data() + e);
        int w = b;
        for (int p = b; p < e; p++) if (p == b || ai[p] != ai[p - 1]) ai[w++] = ai[p];
        len[v] = w - b;
      }
    });
    CK("  adj: sort + dedupe");
    int w = 0;
    for (int v = 0; v < nf; v++) {
      int b = ap[v];
      if (w != b) std::memmove(ai.data() + w, ai.data() + b, sizeof(int32_t) * len[v]);
      ap[v] = w;
      w += len[v];
    }
    ap[nf] = w;
    ai.resize(w);
  }
  CK("adjacency");
  // nested dissection -- from scratch, or the previous ordering extended by the new vertices
  const int NT = host_threads();
  std::vector<int32_t> order(nf), panel_start;
  std::vector<std::pair<int, int>> pos_ranges;
  for (int v = 0; v < nf; v++) order[v] = v;
  std::vector<uint8_t> pstart(nf, 0);
  bool extended = false;
  if (prev && prev->nd_root >= 0 && prev->nf > 0 && prev->nf <= nf && prev->nV <= nV && !fixed) {
    // the old vertices must keep their block indices: every old vertex that has an edge now had one before
    bool same = true;
    for (int v = 0; v < prev->nV && same; v++) same = (S.hidx[v] == prev->hidx[v]);
    static const bool extend_on = !(getenv("CGMR_SYM_EXTEND") && atoi(getenv("CGMR_SYM_EXTEND")) == 0);
    const int n_new = nf - prev->nf;
    // edges [0, n_common) are the previous list's first n_common (default: all of it -- a grown list); the others are checked
    // against the tree one by one; edges the previous list had beyond that and this one has not are simply gone (a tree
    // that separates a graph separates every graph with fewer edges)
    const int nc = n_common < 0 ? prev->nE : std::min(n_common, std::min(prev->nE, nE));
    if (same && extend_on && (n_common >= 0 || prev->nE <= nE) && prev->nd_appended + n_new <= std::max(64, prev->nd_nf_full / 4)) {
      std::vector<int32_t> fh;
      for (int k = 0; k < n_hub_vertices; k++) {
        const int v = hub_vertices[k];
        if (v >= 0 && v < nV && S.hidx[v] >= 0 && std::find(fh.begin(), fh.end(), S.hidx[v]) == fh.end()) fh.push_back(S.hidx[v]);
      }
      extended = extend_order(*prev, nf, ap, ai, order, pstart, pos_ranges, S, nE - nc, ef + nc, et + nc, S.hidx, fh);
    }
  }
  if (!extended) {
    NDCtx C{ap, ai, order, pstart};
    C.vs.assign(nf, NDCtx::VState{0, 0});
    C.queue.assign(nf, 0);
    C.tmp.assign(nf, 0);
    C.recs.assign(2 * (size_t)nf + 4, NDCtx::Rec());
    C.max_par_depth = NT >= 8 ? 3 : (NT >= 4 ? 2 : (NT >= 2 ? 1 : 0));
    { static const int pd = getenv("CGMR_ND_PAR_DEPTH") ? atoi(getenv("CGMR_ND_PAR_DEPTH")) : -1; if (pd >= 0 && NT >= 2) C.max_par_depth = pd; }
    // Hubs -- vertices with far more neighbours than a pose has, i.e. the gauge vertices of the condensed stars received from
    // the peers (30-60 edges each) -- are kept out of the dissection and eliminated last, as one more piece of the root's
    // separator: left inside, a star ties the subtrees its ends lie in together, level by level (robot 0 of the eight-robot
    // C5 rounds: 38 tree levels and 95 MFLOP with them inside, 10-11 levels and 47 MFLOP with the own edges alone).  The
    // sweeps never enter a vertex outside the range being ordered, so leaving the hubs out of the range is all it takes.
    std::vector<int32_t> hubs;
    {
      static const int hub_min = getenv("CGMR_HUB_DEGREE") ? atoi(getenv("CGMR_HUB_DEGREE")) : 32;
      const int thr = std::max(hub_min, (int)(4 * (ai.size() / (size_t)std::max(nf, 1))));
      if (hub_min > 0)
        for (int v = 0; v < nf; v++) if (ap[v + 1] - ap[v] > thr) hubs.push_back(v);
      if ((int)hubs.size() > 3 * kPanelW) {                      // (a graph that is dense all over has no hubs)
        std::sort(hubs.begin(), hubs.end(), [&](int a, int b) { const int da = ap[a + 1] - ap[a], db = ap[b + 1] - ap[b]; return da != db ? da > db : a < b; });
        hubs.resize(3 * kPanelW);
        std::sort(hubs.begin(), hubs.end());
      }
      // ... and the vertices the caller names (a star of fewer than `thr` edges ties subtrees together just the same: the
      // eight-robot C5 rounds had trees of 13-31 levels on 2100 vertices where the own edges alone give 9-10)
      if (hub_min > 0 && n_hub_vertices > 0) {
        for (int k = 0; k < n_hub_vertices; k++) {
          const int v = hub_vertices[k];
          if (v >= 0 && v < nV && S.hidx[v] >= 0) hubs.push_back(S.hidx[v]);
        }
        std::sort(hubs.begin(), hubs.end());
        hubs.erase(std::unique(hubs.begin(), hubs.end()), hubs.end());
      }
      if (!hubs.empty()) {
        std::vector<uint8_t> is_hub(nf, 0);
        for (int v : hubs) is_hub[v] = 1;
        int k = 0;
        for (int v = 0; v < nf; v++) if (!is_hub[v]) order[k++] = v;
        for (int v : hubs) order[k++] = v;
      }
    }
    const int n_nd = nf - (int)hubs.size();
    int root_rec = -1;
    // The root looks for a far vertex first and sweeps from there, like every node below it.  Sweeping from the graph's first
    // pose instead (the start of the odometry chain; one whole-graph BFS less, 0.2 ms) was measured twice: round 3, and round 5
    // over 24 graphs (5k / 10k / 20k poses, eight seeds each): 561 tree levels in all against 523, single graphs 11 levels
    // taller or 12 shorter -- the benchmark graph happens to gain one (18 -> 17 levels, 5.10 -> 4.77 ms device).  Not taken; nor
    // both orderings side by side with the shorter tree kept (500 levels in all, +0.4-0.6 ms of analysis on the shared host
    // for 0.24 ms per level and optimize(10)), nor a choice after three levels of both (517 levels).  DESIGN.md 2.1.
    const NDRange whole = nd(C, 0, n_nd, 0, -1, &root_rec);
    CK("  nd: dissection");
    S.nd_height_full = whole.height + (nf - n_nd + kPanelW - 1) / kPanelW;
    for (int p = n_nd; p < nf; p += kPanelW) pstart[p] = 1;
    pos_ranges.swap(C.subtree_ranges);
    // the tree for the next extension: as recorded (materialize_nd_tree makes the node table when an extension asks for it)
    S.nd_nodes.clear();
    {
      const int nrec = std::min((int)C.recs.size(), C.n_recs.load());
      S.nd_pending_recs.resize(nrec);
      for (int q = 0; q < nrec; q++) S.nd_pending_recs[q] = {C.recs[q].a, C.recs[q].b, C.recs[q].n, C.recs[q].ssz};
      S.nd_pending_order = order;
      S.nd_pending_hubs = hubs;
      S.nd_pending_root = root_rec;
      S.nd_pending = true;
      S.nd_root = (root_rec >= 0 || !hubs.empty()) ? 0 : -1;      // (the root is the first node of the table)
    }
    S.nd_nf_full = nf;
    S.nd_appended = 0;
    CK("  nd: tree recorded");
  }
  S.extended = extended;
  for (int p = 0; p < nf; p++) if (pstart[p]) panel_start.push_back(p);
  std::vector<int32_t> iperm(nf);
  for (int p = 0; p < nf; p++) iperm[order[p]] = p;
  S.perm.assign(nf, -1);
  for (int v = 0; v < nV; v++) if (S.hidx[v] >= 0) { S.vperm[v] = iperm[S.hidx[v]]; S.perm[S.vperm[v]] = v; }
  CK("nested dissection + perm");
  S.t_order = now_s() - t0;
  double t1 = now_s();

  // adjacency in permuted indices, rows sorted
  std::vector<int32_t> cp(nf + 1, 0), ci(ai.size());
  std::vector<int32_t> offbase(nf + 1, 0);             // (the rows' entries below the diagonal are counted in the same pass: see below)
  for (int c = 0; c < nf; c++) cp[c + 1] = cp[c] + (ap[order[c] + 1] - ap[order[c]]);
  parallel_for(nf, NT, [&](int lo, int hi) {
    for (int c = lo; c < hi; c++) {
      int o = order[c], w = cp[c], cnt = 0;
      for (int p = ap[o]; p < ap[o + 1]; p++) { const int r = iperm[ai[p]]; ci[w++] = r; cnt += r > c ? 1 : 0; }
      sort_row(ci.data() + cp[c], ci.data() + cp[c + 1]);
      offbase[c + 1] = cnt;
    }
  });
  CK("permuted adjacency");
  // unique lower off-diagonal blocks: enumerate (c, r>c) column-major
  for (int c = 0; c < nf; c++) offbase[c + 1] += offbase[c];
  S.nb = offbase[nf];
  S.off_row.resize(S.nb);
  S.off_col.resize(S.nb);
  parallel_for(nf, NT, [&](int lo, int hi) {
    for (int c = lo; c < hi; c++) {
      int k = offbase[c];
      for (int p = cp[c]; p < cp[c + 1]; p++) if (ci[p] > c) { S.off_row[k] = ci[p]; S.off_col[k] = c; k++; }
    }
  });
  auto off_id = [&](int r, int c) {   // r > c: the blocks of column c are listed by ascending row, a handful of them
    int k = offbase[c];
    while (S.off_row[k] != r) k++;
    return k;
  };
  CK("off-diagonal blocks");
  // assembly CSR: block -> contributing edge terms
  // A caller with a device takes over here (round 6): everything the lists depend on is final, the device builds them
  // underneath the borders / maps below (AnalyzeHooks::blocks_ready; 0.25-0.35 ms of the eight threads otherwise)
  if (hooks && hooks->blocks_ready) {
    const int hrc = hooks->blocks_ready(S, offbase.data());
    if (hrc) return hrc;
    S.asm_on_device = true;
    S.maps_on_device = hooks->maps_on_device;
    CK("assembly lists: handed to the device");
  }
  if (!S.asm_on_device) {
  // (every thread walks all edges in order and keeps the blocks of its own key range: lists stay in edge order)
  const int nkeys = nf + S.nb;
  S.asm_ptr.assign(nkeys + 1, 0);
  std::vector<int32_t> e_a(nE), e_b(nE), e_off(nE, -1);
  parallel_for(nE, NT, [&](int lo, int hi) {
    for (int k = lo; k < hi; k++) {
      int a = S.vperm[ef[k]], b = S.vperm[et[k]];
      e_a[k] = a;
      e_b[k] = (b >= 0 && a == b) ? -1 : b;              // self edge: one diagonal contribution only
      if (a >= 0 && b >= 0 && a != b) e_off[k] = nf + (a > b ? off_id(a, b) : off_id(b, a));
    }
  });
  CK("  asm: edge keys");
  const int nkt = (nE >= 4096) ? NT : 1;
  auto key_range = [&](int t, int& klo, int& khi) { klo = (int)((int64_t)nkeys * t / nkt); khi = (int)((int64_t)nkeys * (t + 1) / nkt); };
  auto run_keyed = [&](auto&& body) {
    std::vector<HelperPool::Job> jobs(nkt > 1 ? nkt - 1 : 0);
    for (int t = 1; t < nkt; t++) { jobs[t - 1].fn = [&body, t] { body(t); }; pool().run(jobs[t - 1]); }
    body(0);
    for (auto& j : jobs) HelperPool::wait(j);
  };
  run_keyed([&](int t) {
    int klo, khi;
    key_range(t, klo, khi);
    for (int k = 0; k < nE; k++) {
      const int a = e_a[k], b = e_b[k], e = e_off[k];
      if (a >= klo && a < khi) S.asm_ptr[a + 1]++;
      if (b >= klo && b < khi) S.asm_ptr[b + 1]++;
      if (e >= klo && e < khi) S.asm_ptr[e + 1]++;
    }
  });
  CK("  asm: count");
  for (int q = 0; q < nkeys; q++) S.asm_ptr[q + 1] += S.asm_ptr[q];
  S.asm_src.resize(S.asm_ptr[nkeys]);
  CK("  asm: prefix + resize");
  {
    std::vector<int32_t> pos(S.asm_ptr.begin(), S.asm_ptr.end() - 1);
    run_keyed([&](int t) {
      int klo, khi;
      key_range(t, klo, khi);
      for (int k = 0; k < nE; k++) {
        const int a = e_a[k], b = e_b[k], e = e_off[k];
        if (a >= klo && a < khi) S.asm_src[pos[a]++] = 4 * k + 0;
        if (b >= klo && b < khi) S.asm_src[pos[b]++] = 4 * k + 1;
        // a > b: lower block (row a = i, col b = j) is Hij as is; else (row b = j, col a = i) = Hij^T
        if (e >= klo && e < khi) S.asm_src[pos[e]++] = 4 * k + (a > b ? 2 : 3);
      }
    });
  }
  CK("assembly lists");
  }
  // fronts
  int nfr = (int)panel_start.size();
  S.fronts.assign(nfr, FrontDesc());
  S.col_front.assign(nf, 0);
  for (int f = 0; f < nfr; f++) {
    int c0 = panel_start[f], c1 = (f + 1 < nfr) ? panel_start[f + 1] : nf;
    S.fronts[f].c0 = c0;
    S.fronts[f].nc = c1 - c0;
    S.fronts[f].parent = -1;
    S.fronts[f].level = 0;
    for (int c = c0; c < c1; c++) S.col_front[c] = f;
  }
  // border structure, bottom-up (children always have smaller ids than parents).  The halves of the ND nodes at depth
  // kRangeDepth are complete, mutually independent subtrees: their fronts are done in parallel (own row buffer, own
  // stamps; a child whose parent lies outside the range is handed over afterwards), then the few fronts above them
  // in order.  Same result as the one-pass loop.
  std::vector<std::vector<int32_t>> kids(nfr);
  S.rows.clear();
  std::vector<uint8_t> dead(nfr, 0);
  static const bool amalgamate = !(getenv("CGMR_AMALGAMATE") && atoi(getenv("CGMR_AMALGAMATE")) == 0);
  struct BorderCtx {
    std::vector<int32_t> stamp, list;
    std::vector<int32_t>* rows = nullptr;    // row lists of the fronts handled in this context
    int flo = 0, fhi = 0;                   // front range of the context (parents outside are deferred)
    int ndead = 0, max_ns = 0;
  };
  auto do_front = [&](int f, BorderCtx& X) {
    FrontDesc& F = S.fronts[f];
    std::vector<int32_t>& rows = *X.rows;
    int last = F.c0 + F.nc - 1;
    std::vector<int32_t>& list = X.list;
    list.clear();
    for (int c = F.c0; c <= last; c++)
      for (int p = cp[c + 1] - 1; p >= cp[c] && ci[p] > last; p--)
        if (X.stamp[ci[p]] != f) { X.stamp[ci[p]] = f; list.push_back(ci[p]); }
    for (int ch : kids[f]) {
      const FrontDesc& G = S.fronts[ch];
      for (int q = 0; q < G.ns; q++) {
        int r = rows[G.rows_off + q];
        if (r > last && X.stamp[r] != f) { X.stamp[r] = f; list.push_back(r); }
      }
    }
    std::sort(list.begin(), list.end());
    F.rows_off = (int)rows.size();
    F.ns = (int)list.size();
    rows.insert(rows.end(), list.begin(), list.end());
    if (F.ns > 0) {
      int p = S.col_front[list[0]];
      F.parent = p;
      if (p < X.fhi) kids[p].push_back(f);          // (a parent outside the range gets its children afterwards)
    }
    X.max_ns = std::max(X.max_ns, F.ns);
    // Amalgamation: the front just before this one in the elimination order is absorbed when it is a child of this
    // front and the two together still fit one panel.  Its border beyond my columns is part of my border already, so
    // the row list stands; its columns merely carry explicit zeros in the rows it did not reach.  Every merge takes a
    // level out of the paths through it -- the factorisation pays per level, not per flop.
    while (amalgamate && f > X.flo) {
      int g = f - 1;
      while (g >= X.flo && dead[g]) g--;
      if (g < X.flo) break;
      FrontDesc& G = S.fronts[g];
      if (G.parent != f || G.c0 + G.nc != F.c0 || G.nc + F.nc > kPanelW) break;
      std::vector<int32_t>& kf = kids[f];
      auto it = std::find(kf.begin(), kf.end(), g);
      const size_t at = it - kf.begin();
      kf.erase(it);
      kf.insert(kf.begin() + at, kids[g].begin(), kids[g].end());
      for (int ch : kids[g]) S.fronts[ch].parent = f;
      kids[g].clear();
      F.c0 = G.c0;
      F.nc += G.nc;
      for (int c = G.c0; c < G.c0 + G.nc; c++) S.col_front[c] = f;
      dead[g] = 1;
      X.ndead++;
    }
  };
  // front ranges of the independent subtrees
  std::sort(pos_ranges.begin(), pos_ranges.end());
  std::vector<std::pair<int, int>> franges;
  if (NT > 1 && nfr >= 256)
    for (auto& pr : pos_ranges) {
      int flo = (int)(std::lower_bound(panel_start.begin(), panel_start.end(), pr.first) - panel_start.begin());
      int fhi = (int)(std::lower_bound(panel_start.begin(), panel_start.end(), pr.second) - panel_start.begin());
      if (fhi > flo) franges.emplace_back(flo, fhi);
    }
  const int nrange = (int)franges.size();
  std::vector<BorderCtx> rctx(nrange);
  std::vector<std::vector<int32_t>> rrows(nrange);
  std::vector<uint8_t> in_range(nfr, 0);
  {
    std::vector<HelperPool::Job> jobs(nrange);
    // the largest subtrees first (twice as many subtrees as threads: the long ones start at once, the short ones fill in)
    std::vector<int> by_size(nrange);
    for (int t = 0; t < nrange; t++) by_size[t] = t;
    std::sort(by_size.begin(), by_size.end(), [&](int a, int b) {
      const int sa = franges[a].second - franges[a].first, sb = franges[b].second - franges[b].first;
      return sa != sb ? sa > sb : a < b;
    });
    for (int q = 0; q < nrange; q++) {
      const int t = by_size[q];
      for (int f = franges[t].first; f < franges[t].second; f++) in_range[f] = 1;
      jobs[t].fn = [&, t] {
        BorderCtx& X = rctx[t];
        X.stamp.assign(nf, -1);
        X.rows = &rrows[t];
        X.flo = franges[t].first;
        X.fhi = franges[t].second;
        for (int f = X.flo; f < X.fhi; f++) do_front(f, X);
      };
      if (q + 1 < nrange) pool().run(jobs[t]); else { jobs[t].fn(); jobs[t].done.store(1); }
    }
    for (int t = 0; t < nrange; t++) HelperPool::wait(jobs[t]);
  }
  CK("  borders: subtrees");
  int ndead = 0;
  for (int t = 0; t < nrange; t++) {            // splice the row lists together, hand the subtree roots to their parents
    const int base = (int)S.rows.size();
    S.rows.insert(S.rows.end(), rrows[t].begin(), rrows[t].end());
    for (int f = franges[t].first; f < franges[t].second; f++) {
      if (dead[f]) continue;
      FrontDesc& F = S.fronts[f];
      F.rows_off += base;
      if (F.ns > 0 && F.parent >= franges[t].second) kids[F.parent].push_back(f);
    }
    ndead += rctx[t].ndead;
    S.max_ns = std::max(S.max_ns, rctx[t].max_ns);
  }
  {
    BorderCtx X;                                 // the fronts above the subtrees, in order
    X.stamp.assign(nf, -1);
    X.rows = &S.rows;
    X.flo = 0;
    X.fhi = nfr;
    for (int f = 0; f < nfr; f++) {
      if (in_range[f]) continue;
      std::sort(kids[f].begin(), kids[f].end());   // children in id order, as the one-pass loop lists them
      do_front(f, X);
    }
    ndead += X.ndead;
    S.max_ns = std::max(S.max_ns, X.max_ns);
  }
  CK("  borders: splice + fronts above");
  if (ndead > 0) {                    // compact the front table (ids stay monotone: children before parents)
    std::vector<int32_t> newid(nfr, -1);
    int k = 0;
    for (int f = 0; f < nfr; f++) if (!dead[f]) newid[f] = k++;
    for (int f = 0; f < nfr; f++) {
      if (dead[f]) continue;
      FrontDesc F = S.fronts[f];
      if (F.parent >= 0) F.parent = newid[F.parent];
      S.fronts[newid[f]] = F;
      std::vector<int32_t> kf;
      for (int ch : kids[f]) kf.push_back(newid[ch]);
      kids[newid[f]] = std::move(kf);
    }
    nfr = k;
    S.fronts.resize(nfr);
    kids.resize(nfr);
    for (int c = 0; c < nf; c++) S.col_front[c] = newid[S.col_front[c]];
  }
  // Levels (0 = no children) and the schedule of the children's contributions.  A child adds the leading slab of its
  // update matrix into its parent's assembled panel during ONE update launch t, level(child) <= t < level(parent); two
  // children of a front that share a launch add into different copies of the panel (read-modify-write without
  // atomics: every cell of a copy has one writer per launch, launches are ordered => bit-reproducible sums).  The
  // children with the least slack are placed first, each into the launch with the fewest siblings so far; when more than
  // kMaxPanSlots would share a launch the parent moves up a level.
  CK("  borders: compaction");
  int nlev = 0;
  {
    std::vector<int32_t> cnt, ord;
    for (int f = 0; f < nfr; f++) {
      int lv = 0;
      for (int ch : kids[f]) lv = std::max(lv, S.fronts[ch].level + 1);
      ord.assign(kids[f].begin(), kids[f].end());
      std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return S.fronts[a].level > S.fronts[b].level; });
      for (;;) {
        cnt.assign(lv, 0);
        bool ok = true;
        for (int ch : ord) {
          int best = -1;
          for (int t = S.fronts[ch].level; t < lv; t++) if (best < 0 || cnt[t] < cnt[best]) best = t;
          if (cnt[best] >= kMaxPanSlots) { ok = false; break; }
          S.fronts[ch].sched_t = best;
          S.fronts[ch].sched_slot = cnt[best]++;
        }
        if (ok) break;
        lv++;
      }
      S.fronts[f].pan_slots = 1;
      for (int c : cnt) S.fronts[f].pan_slots = std::max(S.fronts[f].pan_slots, c);
      S.fronts[f].level = lv;
      S.fronts[f].sched_t = lv;          // (roots: no update tiles)
      S.fronts[f].sched_slot = 0;
      nlev = std::max(nlev, lv + 1);
    }
  }
  CK("borders + amalgamation");
  // children lists, rel / inv maps, A lists, offsets: the sizes first (serial, cheap), then the contents in parallel
  int64_t Loff = 0, Uoff = 0, Panoff = 0;
  double flops = 0;
  {
    int64_t n_child = 0, n_rel = 0, n_inv = 0, n_a = 0;
    for (int f = 0; f < nfr; f++) {
      FrontDesc& F = S.fronts[f];
      // children: the big ones first (see slab_is_small), each group in id order
      if (kids[f].size() > 1) {
        const int cend = F.c0 + F.nc;
        std::stable_partition(kids[f].begin(), kids[f].end(), [&](int ch) {
          const FrontDesc& G = S.fronts[ch];
          const int32_t* gr = S.rows.data() + G.rows_off;
          const int na = (int)(std::lower_bound(gr, gr + G.ns, cend) - gr);
          return !slab_is_small(G.ns, na);
        });
      }
      F.child_off = (int)n_child;
      F.nchild = (int)kids[f].size();
      n_child += F.nchild;
      for (int ch : kids[f]) {
        FrontDesc& G = S.fronts[ch];
        G.rel_off = (int)n_rel; n_rel += G.ns;
        G.inv_off = (int)n_inv; n_inv += F.ns;
      }
      F.a_off = (int)n_a;
      F.a_cnt = F.nc + (offbase[F.c0 + F.nc] - offbase[F.c0]);
      n_a += F.a_cnt;
      int64_t w = 3 * (int64_t)F.nc, r = 3 * (int64_t)F.ns;
      const int64_t lw = kFrontW;
      F.L_off = Loff; Loff += factor_header((int)lw) + r * lw + lw * lw;   // header, L21, then L11^-1 (the chained backward solve's)
      F.U_off = Uoff; Uoff += (r * r + r + 3) & ~int64_t(1);      // even offsets: the update matrices are read with 16-byte loads
      F.pan_off = Panoff; Panoff += pan_size(F.ns) * F.pan_slots;
      flops += (double)w * w * w / 3.0 + (double)r * w * w + (double)r * r * w;
    }
    S.children.resize(n_child);
    S.n_rel = n_rel; S.n_inv = n_inv;
    if (!S.maps_on_device) {
      S.rel.resize(n_rel);
      S.inv.assign(n_inv, -1);
      S.alist.resize(3 * n_a);
    }
  }
  if (S.maps_on_device) {
    // the device fills rel / inv / blk_dst / b_dst from the uploaded front table (gn_structure.hip: k_build_maps); the host
    // keeps what it reads itself: the children lists and every child's count of border rows inside its parent's own columns
    parallel_for(nfr, NT, [&](int flo, int fhi) {
      for (int f = flo; f < fhi; f++) {
        const FrontDesc& F = S.fronts[f];
        std::copy(kids[f].begin(), kids[f].end(), S.children.begin() + F.child_off);
        for (int ch : kids[f]) {
          FrontDesc& G = S.fronts[ch];
          const int32_t* gr = S.rows.data() + G.rows_off;
          G.na = (int)(std::lower_bound(gr, gr + G.ns, F.c0 + F.nc) - gr);
        }
      }
    }, 256);
  } else
  parallel_for(nfr, NT, [&](int flo, int fhi) {
    std::vector<int32_t> posmap(nf, -1);                 // border row -> position in the current front's row list
    for (int f = flo; f < fhi; f++) {
      const FrontDesc& F = S.fronts[f];
      std::copy(kids[f].begin(), kids[f].end(), S.children.begin() + F.child_off);
      for (int q = 0; q < F.ns; q++) posmap[S.rows[F.rows_off + q]] = q;
      // maps of each child into this front
      for (int ch : kids[f]) {
        FrontDesc& G = S.fronts[ch];
        int32_t* rel = S.rel.data() + G.rel_off;
        int32_t* inv = S.inv.data() + G.inv_off;
        int na = 0;
        for (int q = 0; q < G.ns; q++) {
          int r = S.rows[G.rows_off + q];
          if (r < F.c0 + F.nc) { rel[q] = r - F.c0; na++; }
          else { int p = posmap[r]; rel[q] = F.nc + p; inv[p] = q; }
        }
        G.na = na;
      }
      // A blocks of this front's columns
      int32_t* al = S.alist.data() + 3 * (size_t)F.a_off;
      for (int c = F.c0; c < F.c0 + F.nc; c++) {
        int lc = c - F.c0;
        *al++ = c; *al++ = lc; *al++ = lc;
        int k = offbase[c];
        for (int p = cp[c]; p < cp[c + 1]; p++) {
          int r = ci[p];
          if (r <= c) continue;
          int lr = (r < F.c0 + F.nc) ? r - F.c0 : F.nc + posmap[r];
          *al++ = nf + k; *al++ = lr; *al++ = lc;
          k++;
        }
      }
    }
  }, 256);
  CK("maps + A lists");
  S.L_doubles = Loff;
  S.U_doubles = Uoff;
  S.pan_doubles = Panoff;
  S.flops = flops;
  // levels
  S.level_ptr.assign(nlev + 1, 0);
  for (const FrontDesc& F : S.fronts) S.level_ptr[F.level + 1]++;
  for (int l = 0; l < nlev; l++) S.level_ptr[l + 1] += S.level_ptr[l];
  S.level_fronts.resize(nfr);
  {
    std::vector<int32_t> pos(S.level_ptr.begin(), S.level_ptr.end() - 1);
    for (int f = 0; f < nfr; f++) S.level_fronts[pos[S.fronts[f].level]++] = f;
  }
  // ---- top block: walk down the root's chain while the columns stay consecutive and fit
  S.gn_level_ptr = S.level_ptr;
  S.gn_level_fronts = S.level_fronts;
  static const bool top_on = !(getenv("CGMR_TOP_BLOCK") && atoi(getenv("CGMR_TOP_BLOCK")) == 0);
  if (top_on && nfr > 0) {
    int root = nfr - 1;                                    // the last front of the elimination order
    const FrontDesc& Rt = S.fronts[root];
    if (Rt.parent < 0 && Rt.ns == 0 && Rt.level == nlev - 1 && S.level_ptr[nlev] - S.level_ptr[nlev - 1] == 1 &&
        3 * Rt.nc <= kTopMaxCols) {
      std::vector<int32_t> chain(1, root);
      int cols = 3 * Rt.nc;
      for (;;) {
        const FrontDesc& B = S.fronts[chain.back()];
        int next = -1;
        for (int k = 0; k < B.nchild; k++) {
          const int c = S.children[B.child_off + k];
          const FrontDesc& C = S.fronts[c];
          if (C.c0 + C.nc == B.c0 && C.level == B.level - 1) { next = c; break; }
        }
        if (next < 0) break;
        const FrontDesc& C = S.fronts[next];
        // every border row of the candidate must lie inside the block (it does when the chain ends in a border-less
        // root and the columns are consecutive), its level must use the narrow panels
        if (cols + 3 * C.nc > kTopMaxCols) break;
        chain.push_back(next);
        cols += 3 * C.nc;
      }
      std::reverse(chain.begin(), chain.end());            // bottom-up
      S.top_fronts = chain;
      S.top_c0 = S.fronts[chain[0]].c0;
      S.top_nposes = cols / 3;
      std::vector<uint8_t> in_top(nfr, 0);
      for (int f : chain) in_top[f] = 1;
      for (int f = 0; f < nfr; f++)
        if (!in_top[f] && S.fronts[f].parent >= 0 && in_top[S.fronts[f].parent]) S.top_children.push_back(f);
      for (int f : chain) {
        // the front's A blocks in the order of its alist: per own column the diagonal block, then the column's blocks below
        // the diagonal by ascending row (the global row of a block is all the dense top block needs)
        const FrontDesc& F = S.fronts[f];
        int k = 0;
        for (int c = F.c0; c < F.c0 + F.nc; c++) {
          S.top_blocks.push_back(F.a_off + k++);
          S.top_blocks.push_back(c - S.top_c0);
          S.top_blocks.push_back(c - S.top_c0);
          for (int p = cp[c]; p < cp[c + 1]; p++) {
            if (ci[p] <= c) continue;
            S.top_blocks.push_back(F.a_off + k++);
            S.top_blocks.push_back(ci[p] - S.top_c0);
            S.top_blocks.push_back(c - S.top_c0);
          }
        }
      }
      // Gauss-Newton level lists without the block
      S.gn_level_fronts.clear();
      S.gn_level_ptr.assign(1, 0);
      for (int l = 0; l < nlev; l++) {
        for (int q = S.level_ptr[l]; q < S.level_ptr[l + 1]; q++)
          if (!in_top[S.level_fronts[q]]) S.gn_level_fronts.push_back(S.level_fronts[q]);
        S.gn_level_ptr.push_back((int)S.gn_level_fronts.size());      // (GN level index = tree level: the children's schedule counts in tree levels)
      }
      while (S.gn_level_ptr.size() > 1 && S.gn_level_ptr[S.gn_level_ptr.size() - 1] == S.gn_level_ptr[S.gn_level_ptr.size() - 2])
        S.gn_level_ptr.pop_back();                                     // the block's own levels
    }
  }
  CK("levels + top block");
  // ---- where every front's contribution and every H block goes
  {
    std::vector<uint8_t> in_top(nfr, 0);
    for (int f : S.top_fronts) in_top[f] = 1;
    for (int f = 0; f < nfr; f++) {
      FrontDesc& F = S.fronts[f];
      F.front_id = f;
      F.p_c0 = F.parent >= 0 ? S.fronts[F.parent].c0 : -1;
      F.ppan_off = -1; F.p_nc = 0; F.p_ns = 0;
      if (F.parent >= 0 && !in_top[F.parent]) {
        const FrontDesc& Pf = S.fronts[F.parent];
        F.ppan_off = Pf.pan_off + pan_size(Pf.ns) * F.sched_slot;
        F.p_nc = Pf.nc; F.p_ns = Pf.ns;
      } else {
        F.sched_t = F.level;               // the whole update matrix goes to Ubuf, in the front's own launch
        F.sched_slot = 0;
      }
    }
    if (S.maps_on_device) {
      if (Panoff + kFrontW > 0x7fffffff) return -2;          // (the destinations are 32-bit offsets into the panels)
    } else {
    S.blk_dst.assign((size_t)nf + S.nb, 0);
    S.b_dst.assign(nf, -1);
    for (int f = 0; f < nfr; f++) {
      const FrontDesc& F = S.fronts[f];
      for (int k = 0; k < F.a_cnt; k++) {
        const int32_t* al = S.alist.data() + 3 * (size_t)(F.a_off + k);
        const int blk = al[0], lr = al[1], lc = al[2];
        if (in_top[f]) { S.blk_dst[blk] = -(F.a_off + k + 1); continue; }
        const int64_t row = lr < F.nc ? 3 * lr : kFrontW + 3 * (lr - F.nc);
        const int64_t off = F.pan_off + row * kPanStride + 3 * lc;
        if (off > 0x7fffffff) return -2;                     // (a graph two orders of magnitude beyond the benchmark configurations)
        S.blk_dst[blk] = (int32_t)off;
      }
      if (!in_top[f])
        for (int c = 0; c < F.nc; c++) S.b_dst[F.c0 + c] = (int32_t)(F.pan_off + (int64_t)(kFrontW + 3 * F.ns) * kPanStride + 3 * c);
    }
    }
  }
  CK("destinations");
  S.t_struct = now_s() - t1;
  return 0;
}

}  // namespace cgmr
