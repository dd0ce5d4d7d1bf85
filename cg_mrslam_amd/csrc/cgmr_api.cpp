// C ABI of libcgmr.so (include/cgmr.h): context management and the Gauss-Newton driver.
// Compiled with hipcc; contains no kernels (those live in gn_kernels.hip / matcher_kernels.hip).
#include "cgmr_ctx.h"
#include "gn_host.h"

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace cgmr {

double wall_s() {
  using namespace std::chrono;
  return duration_cast<duration<double>>(steady_clock::now().time_since_epoch()).count();
}

int set_err(cgmr_ctx* ctx, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf;
  return code;
}

#define HIP_TRY(ctx, call)                                                                      \
  do {                                                                                          \
    hipError_t e_ = (call);                                                                     \
    if (e_ != hipSuccess) return set_err(ctx, CGMR_E_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
  } while (0)

int side_stream(cgmr_ctx* ctx) {
  if (ctx->side) return 0;
  // (default priority.  Round 4 tried the lowest priority here and the highest for the context's stream -- the batches as
  // gap fillers of the solve's dependent chain of small launches --: with eight robots on one device a starved batch held
  // up its own robot's next structure upload, optimize(5) 2.57 -> 4.49 ms in the loopback; without peers on the device no
  // gain either way.)
  HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
  HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->side_fork, hipEventDisableTiming));
  HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->side_tail, hipEventDisableTiming));
  return 0;
}

int side_fork(cgmr_ctx* ctx) {
  int rc = side_stream(ctx);
  if (rc) return rc;
  HIP_TRY(ctx, hipEventRecord(ctx->side_fork, ctx->stream));
  HIP_TRY(ctx, hipStreamWaitEvent(ctx->side, ctx->side_fork, 0));
  return 0;
}

int side_mark(cgmr_ctx* ctx) {
  HIP_TRY(ctx, hipEventRecord(ctx->side_tail, ctx->side));
  ctx->side_busy = true;
  return 0;
}

int side_join_host(cgmr_ctx* ctx) {
  if (!ctx->side_busy) return 0;
  HIP_TRY(ctx, hipEventSynchronize(ctx->side_tail));
  ctx->side_busy = false;
  return 0;
}

int side_join_stream(cgmr_ctx* ctx, hipStream_t st) {
  if (!ctx->side_busy) return 0;
  HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->side_tail, 0));
  return 0;
}

// An arena that has to grow is not freed on the spot: hipFree waits for the whole device -- every stream of every context
// of the process, e.g. the other robots' batches on their side streams -- and work queued on this context's streams may
// still read the old block.  The old block goes to the context's graveyard (freed with the context); the arena at least
// doubles, so the graveyard never holds more than the live blocks do (sized for 288 GB of HBM, not for thrift).
int arena_reserve(cgmr_ctx* ctx, Arena& A, size_t bytes) {
  if (bytes <= A.cap) return 0;
  if (A.ptr) { ctx->graveyard.push_back({A.ptr, nullptr}); A.ptr = nullptr; }
  size_t want = std::max(bytes + bytes / 2, 2 * A.cap) + (1 << 20);
  A.cap = 0;
  hipError_t e = hipMalloc((void**)&A.ptr, want);
  if (e != hipSuccess) return set_err(ctx, CGMR_E_ALLOC, "hipMalloc(%zu): %s", want, hipGetErrorString(e));
  A.cap = want;
  return 0;
}

int pinned_reserve(cgmr_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->pinned_cap) return 0;
  if (ctx->pinned) { (void)hipStreamSynchronize(ctx->stream); (void)side_join_host(ctx); (void)hipHostFree(ctx->pinned); ctx->pinned = nullptr; ctx->pinned_cap = 0; }
  size_t want = std::max(2 * bytes, (size_t)4 << 20);      // (page-locking is slow and a robot's graph grows every round: double, never less than 4 MB)
  hipError_t e = hipHostMalloc((void**)&ctx->pinned, want, hipHostMallocDefault);
  if (e != hipSuccess) return set_err(ctx, CGMR_E_ALLOC, "hipHostMalloc(%zu): %s", want, hipGetErrorString(e));
  ctx->pinned_cap = want;
  return 0;
}

int pinned_mask_reserve(cgmr_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->pinned_mask_cap) return 0;
  if (ctx->pinned_mask) { (void)hipStreamSynchronize(ctx->stream); (void)side_join_host(ctx); (void)hipHostFree(ctx->pinned_mask); ctx->pinned_mask = nullptr; ctx->pinned_mask_cap = 0; }
  size_t want = bytes + bytes / 2 + 4096;
  hipError_t e = hipHostMalloc((void**)&ctx->pinned_mask, want, hipHostMallocDefault);
  if (e != hipSuccess) return set_err(ctx, CGMR_E_ALLOC, "hipHostMalloc(%zu): %s", want, hipGetErrorString(e));
  ctx->pinned_mask_cap = want;
  return 0;
}

struct BlobLayout {
  size_t off = 0;
  template <typename T>
  size_t add(size_t count) {
    off = (off + 15) & ~size_t(15);
    size_t o = off;
    off += count * sizeof(T);
    return o;
  }
};

// AnalyzeHooks::blocks_ready of a context's analysis: what the assembly lists depend on goes to the device as soon as it is
// final (vperm, the edge list, the off-diagonal blocks' rows / columns / column starts: one copy from a pinned block of its own)
// and the device builds the lists (gn_structure.hip) underneath the rest of the host's analysis.  CGMR_ASM_DEVICE=0: the host
// builds them as before and they travel with the structure blob.
int gn_upload_early(cgmr_ctx* ctx, const Symbolic& S, const int32_t* ef, const int32_t* et, const int32_t* offbase) {
  const size_t nV = (size_t)S.nV, nE = (size_t)S.nE, nb = (size_t)S.nb, nf = (size_t)S.nf, nkeys = nf + nb;
  BlobLayout B;
  const size_t o_vperm = B.add<int32_t>(nV), o_ef = B.add<int32_t>(nE), o_et = B.add<int32_t>(nE), o_orow = B.add<int32_t>(nb),
               o_ocol = B.add<int32_t>(nb), o_obase = B.add<int32_t>(nf + 1);
  const size_t up_bytes = (B.off + 255) & ~size_t(255);
  B.off = up_bytes;
  const size_t o_asmp = B.add<int32_t>(nkeys + 1), o_asms = B.add<int32_t>(3 * nE + 4), o_cnt = B.add<int32_t>(nkeys + 4),
               o_ekey = B.add<int32_t>(nE), o_long = B.add<int32_t>(3 * nE / 16 + 2), o_tmp = B.add<int32_t>(3 * nE + 4);
  // whatever still reads the previous structure on the side stream (a batch of condensed-graph passes) comes first
  int rc = side_join_stream(ctx, ctx->stream);
  if (rc) return rc;
  rc = arena_reserve(ctx, ctx->st_arena, B.off + 256);
  if (rc) return rc;
  if (!ctx->ev_st_copied) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_st_copied, hipEventDisableTiming));
  else HIP_TRY(ctx, hipEventSynchronize(ctx->ev_st_copied));   // (the previous copy out of the staging block: long done)
  if (up_bytes > ctx->pinned_st_cap) {
    if (ctx->pinned_st) { (void)hipHostFree(ctx->pinned_st); ctx->pinned_st = nullptr; ctx->pinned_st_cap = 0; }
    // (page-locking is slow and a robot's graph grows every round: double, and never less than 2 MB)
    const size_t want = std::max(2 * up_bytes, (size_t)2 << 20);
    hipError_t e = hipHostMalloc((void**)&ctx->pinned_st, want, hipHostMallocDefault);
    if (e != hipSuccess) return set_err(ctx, CGMR_E_ALLOC, "hipHostMalloc(%zu): %s", want, hipGetErrorString(e));
    ctx->pinned_st_cap = want;
  }
  char* h = ctx->pinned_st;
  host_run_tasks(4, [&](int task) {
    switch (task) {
      case 0: memcpy(h + o_ef, ef, 4 * nE); break;
      case 1: memcpy(h + o_et, et, 4 * nE); break;
      case 2: memcpy(h + o_orow, S.off_row.data(), 4 * nb); memcpy(h + o_obase, offbase, 4 * (nf + 1)); break;
      default: memcpy(h + o_ocol, S.off_col.data(), 4 * nb); memcpy(h + o_vperm, S.vperm.data(), 4 * nV); break;
    }
  });
  char* d = ctx->st_arena.ptr;
  HIP_TRY(ctx, hipMemcpyAsync(d, h, up_bytes, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipEventRecord(ctx->ev_st_copied, ctx->stream));
  cgmr_ctx::StView& V = ctx->st_view;
  V.vperm = (int32_t*)(d + o_vperm); V.ef = (int32_t*)(d + o_ef); V.et = (int32_t*)(d + o_et);
  V.off_row = (int32_t*)(d + o_orow); V.off_col = (int32_t*)(d + o_ocol); V.offbase = (int32_t*)(d + o_obase);
  V.asm_ptr = (int32_t*)(d + o_asmp); V.asm_src = (int32_t*)(d + o_asms);
  AsmBuild A;
  A.nE = S.nE; A.nf = S.nf; A.nb = S.nb;
  A.vperm = V.vperm; A.ef = V.ef; A.et = V.et; A.off_row = V.off_row; A.offbase = (const int32_t*)(d + o_obase);
  A.asm_ptr = V.asm_ptr; A.asm_src = V.asm_src; A.ekey = (int32_t*)(d + o_ekey); A.cnt = (int32_t*)(d + o_cnt);
  A.longlist = (int32_t*)(d + o_long); A.tmp = (int32_t*)(d + o_tmp);
  launch_build_asm(ctx->stream, A);
  HIP_TRY(ctx, hipGetLastError());
  return 0;
}

// Lay out and upload the structure arrays; point GnDevice into the arena.
int gn_upload(cgmr_ctx* ctx, const Symbolic& S, const int32_t* ef, const int32_t* et, int iters) {
  // the factor kernels keep row positions (own columns + border rows) in 16-bit LDS maps
  if (3 * (int64_t)S.max_ns + kFrontW > 32767)
    return set_err(ctx, CGMR_E_INVALID, "a front has %d border poses; at most %d are supported", S.max_ns, (32767 - kFrontW) / 3);
  GnDevice& D = ctx->gn;
  D.nV = S.nV; D.nE = S.nE; D.nf = S.nf; D.nb = S.nb;
  D.nfronts = (int)S.fronts.size();
  D.nlevels = (int)S.gn_level_ptr.size() - 1;           // Gauss-Newton levels: without the top block
  D.h_level_ptr = S.gn_level_ptr;
  D.nlevels_full = (int)S.level_ptr.size() - 1;
  D.h_flevel_ptr = S.level_ptr;
  const std::vector<int32_t>& LF = S.gn_level_fronts;
  static const int mid_chunk = getenv("CGMR_CHUNK") ? std::min(kChunkRows, std::max(16, atoi(getenv("CGMR_CHUNK")))) : kMidChunkRows;
  static const int leaf_chunk = getenv("CGMR_LEAF_CHUNK") ? atoi(getenv("CGMR_LEAF_CHUNK")) : kLeafChunkRows;
  // sizing pass: work records (one per front and row chunk) and update tiles per level fix the blob layout; the records
  // themselves are written straight into the pinned staging blob further down, by several host threads
  D.h_tile_ptr.assign(D.nlevels + 1, 0);
  D.h_work_ptr.assign(D.nlevels + 1, 0);
  D.h_level_chrows.assign(D.nlevels, 1);
  D.h_level_leaf.assign(D.nlevels, 1);
  D.h_level_chunk.assign(D.nlevels, kChunkRows);
  D.h_level_mergeable.assign(D.nlevels, 1);
  // update tiles run in the launch the schedule gave their front (sched_t: its own level or, with slack, a later one)
  std::vector<std::vector<int32_t>> sched(D.nlevels);
  for (int q = 0; q < (int)LF.size(); q++) {
    const FrontDesc& F = S.fronts[LF[q]];
    if (F.ns > 0) sched[std::min(F.sched_t, D.nlevels - 1)].push_back(LF[q]);
  }
  for (auto& v : sched) std::sort(v.begin(), v.end());
  std::vector<int32_t> rec0_of(S.fronts.size(), -1);        // a front's first work record: the update tiles address the front through it
  for (int l = 0; l < D.nlevels; l++) {
    for (int q = S.gn_level_ptr[l]; q < S.gn_level_ptr[l + 1]; q++)
      if (S.fronts[LF[q]].nchild > 0) D.h_level_leaf[l] = 0;
    // a level of leaves has its own chunk length (kLeafChunkRows)
    int chunk_rows = (D.h_level_leaf[l] ? std::min(kChunkRows, std::max(16, leaf_chunk)) : mid_chunk);
    // the few fronts of an upper level (an idle chip): shorter chunks = more workgroups per front, each with less of the
    // panel to pull in on its own (a lone workgroup fetches cold data at 10-25 bytes per clock) -- and few extra records
    static const int top_chunk = getenv("CGMR_TOP_CHUNK") ? std::min(kChunkRows, std::max(16, atoi(getenv("CGMR_TOP_CHUNK")))) : kTopChunkRows;
    static const int top_fronts = getenv("CGMR_TOP_CHUNK_FRONTS") ? atoi(getenv("CGMR_TOP_CHUNK_FRONTS")) : kTopChunkFronts;
    if (!D.h_level_leaf[l] && S.gn_level_ptr[l + 1] - S.gn_level_ptr[l] <= top_fronts) chunk_rows = std::min(chunk_rows, top_chunk);
    D.h_level_chunk[l] = chunk_rows;
    int nwork = 0;
    for (int q = S.gn_level_ptr[l]; q < S.gn_level_ptr[l + 1]; q++) {
      const int r = 3 * S.fronts[LF[q]].ns;
      rec0_of[LF[q]] = D.h_work_ptr[l] + nwork;
      nwork += std::max(1, (r + chunk_rows - 1) / chunk_rows);
      D.h_level_chrows[l] = std::max(D.h_level_chrows[l], std::min(r, chunk_rows) + 1);
    }
    int xload[8] = {0, 0, 0, 0, 0, 0, 0, 0};                     // update tiles per XCD (see the tile list below)
    for (int f : sched[l]) {
      if (S.fronts[f].nchild > kWorkChildren) D.h_level_mergeable[l] = 0;
      const int T = (3 * S.fronts[f].ns + 31) / 32;
      *std::min_element(xload, xload + 8) += T * (T + 1) / 2;
    }
    D.h_tile_ptr[l + 1] = D.h_tile_ptr[l] + 8 * *std::max_element(xload, xload + 8);
    D.h_work_ptr[l + 1] = D.h_work_ptr[l] + nwork;
  }
  const size_t n_tiles = (size_t)D.h_tile_ptr[D.nlevels], n_work = (size_t)D.h_work_ptr[D.nlevels];
  BlobLayout B;
  size_t o_fronts = B.add<FrontDesc>(S.fronts.size());
  size_t o_fronts_lv = B.add<FrontDesc>(S.fronts.size());   // the same descriptors in level order: the solves index them by workgroup
  size_t o_rows = B.add<int32_t>(S.rows.size());
  size_t o_children = B.add<int32_t>(S.children.size());
  const bool dev_maps = S.maps_on_device;                   // rel / inv / blk_dst / b_dst are made on the device (below): no staging, no upload
  size_t o_rel = B.add<int32_t>(dev_maps ? 0 : S.rel.size());
  size_t o_inv = B.add<int32_t>(dev_maps ? 0 : S.inv.size());
  size_t o_bdst = B.add<int32_t>(dev_maps ? 0 : S.blk_dst.size());
  size_t o_rdst = B.add<int32_t>(dev_maps ? 0 : S.b_dst.size());
  size_t o_lf = B.add<int32_t>(S.level_fronts.size());
  size_t o_tiles = B.add<int32_t>(3 * n_tiles);
  size_t o_work = B.add<WorkRec>(dev_maps ? 0 : n_work);     // (device-made with the maps: k_build_maps)
  size_t o_rec0 = B.add<int32_t>(dev_maps ? 2 * S.fronts.size() : 0);
  // (S.asm_on_device: these seven are on the device already, gn_upload_early)
  const bool early = S.asm_on_device;
  size_t o_asmp = B.add<int32_t>(early ? 0 : S.asm_ptr.size());
  size_t o_asms = B.add<int32_t>(early ? 0 : S.asm_src.size());
  size_t o_vperm = B.add<int32_t>(early ? 0 : S.vperm.size());
  size_t o_ef = B.add<int32_t>(early ? 0 : S.nE);
  size_t o_et = B.add<int32_t>(early ? 0 : S.nE);
  size_t o_orow = B.add<int32_t>(early ? 0 : S.off_row.size());
  size_t o_ocol = B.add<int32_t>(early ? 0 : S.off_col.size());
  size_t o_tf = B.add<int32_t>(S.top_fronts.size()), o_tc = B.add<int32_t>(S.top_children.size()), o_tb = B.add<int32_t>(S.top_blocks.size());
  size_t blob_bytes = (B.off + 255) & ~size_t(255);
  // numeric work space
  BlobLayout N;
  N.off = blob_bytes;
  size_t o_term = N.add<double>((size_t)34 * S.nE);
  size_t o_A = N.add<double>((size_t)9 * (S.nf + S.nb));
  size_t o_b = N.add<double>((size_t)3 * S.nf);
  size_t o_y = N.add<double>((size_t)3 * S.nf);
  size_t o_x = N.add<double>((size_t)3 * S.nf);
  size_t o_u = N.add<double>((size_t)3 * S.rows.size() + 3);
  size_t o_L = N.add<double>((size_t)S.L_doubles + 1);
  size_t o_U = N.add<double>((size_t)S.U_doubles + 1);
  size_t o_pan = N.add<double>((size_t)S.pan_doubles + 2);
  size_t o_chi = N.add<double>((size_t)iters + 2);
  size_t o_status = N.add<int>(4);
  size_t o_ready = N.add<int>(S.fronts.size() + 4);
  size_t o_cmask = N.add<uint8_t>((size_t)S.nf + 16);
  if (dev_maps) {
    o_rel = N.add<int32_t>((size_t)S.n_rel + 4); o_inv = N.add<int32_t>((size_t)S.n_inv + 4);
    o_bdst = N.add<int32_t>((size_t)S.nf + S.nb + 4); o_rdst = N.add<int32_t>((size_t)S.nf + 4);
    o_work = N.add<WorkRec>(n_work + 1);
  }
  size_t total = N.off + 256;
  int rc = arena_reserve(ctx, ctx->gn_arena, total);
  if (rc) return rc;
  rc = pinned_reserve(ctx, blob_bytes);
  if (rc) return rc;
  char* h = ctx->pinned;
  auto put = [&](size_t off, const void* src, size_t bytes) { if (bytes) memcpy(h + off, src, bytes); };
  // staging: independent pieces on the analysis' helper threads
  host_run_tasks(6, [&](int task) {
    switch (task) {
      case 0: {                                                  // work records + update tiles, level by level
        WorkRec* work = dev_maps ? nullptr : reinterpret_cast<WorkRec*>(h + o_work);
        int32_t* tiles = reinterpret_cast<int32_t*>(h + o_tiles);
        if (dev_maps) {                                          // the device writes the records: where each front's begin, how many
          int32_t* r0 = reinterpret_cast<int32_t*>(h + o_rec0);
          for (size_t f = 0; f < S.fronts.size(); f++) { r0[2 * f] = -1; r0[2 * f + 1] = 0; }
        }
        // Update tiles of one front sit 8 apart in the launch: workgroup b runs on XCD b % 8 (observed; speed only), so
        // the tiles that share the front's L21 rows share one L2 instead of fetching them into up to eight.  A front goes
        // to the XCD with the fewest tiles so far; the shorter queues are padded with empty entries (rec = -1).
        for (int l = 0; l < D.nlevels; l++) {
          const int chunk_rows = D.h_level_chunk[l];
          int w = D.h_work_ptr[l];
          for (int q = S.gn_level_ptr[l]; q < S.gn_level_ptr[l + 1]; q++) {
            const int f = LF[q];
            const int r = 3 * S.fronts[f].ns;
            const int nchunk = std::max(1, (r + chunk_rows - 1) / chunk_rows);
            if (dev_maps) {
              int32_t* r0 = reinterpret_cast<int32_t*>(h + o_rec0);
              r0[2 * f] = w; r0[2 * f + 1] = nchunk;
              w += nchunk;
              continue;
            }
            for (int c = 0; c < nchunk; c++) {
              WorkRec& wr = work[w++];
              memset(&wr, 0, sizeof wr);
              wr.F = S.fronts[f];
              wr.front = f;
              wr.chunk = c;
              for (int k = 0; k < std::min<int>(wr.F.nchild, kWorkChildren); k++) {
                const FrontDesc& G = S.fronts[S.children[wr.F.child_off + k]];
                WorkChild& wc = wr.ch[k];
                wc.U_off = G.U_off; wc.ns = G.ns; wc.na = G.na;
                wc.rel_off = G.rel_off; wc.inv_off = G.inv_off; wc.rows_off = G.rows_off;
              }
            }
          }
          int32_t* tl = tiles + 3 * (size_t)D.h_tile_ptr[l];
          const int slots = D.h_tile_ptr[l + 1] - D.h_tile_ptr[l];
          for (int k = 0; k < slots; k++) { tl[3 * k] = -1; tl[3 * k + 1] = 0; tl[3 * k + 2] = 0; }
          int xload[8] = {0, 0, 0, 0, 0, 0, 0, 0};
          for (int f : sched[l]) {
            const int T = (3 * S.fronts[f].ns + 31) / 32;
            const int x = (int)(std::min_element(xload, xload + 8) - xload);
            for (int ti = 0; ti < T; ti++)
              for (int tj = 0; tj <= ti; tj++) {
                int32_t* e = tl + 3 * (size_t)(8 * xload[x]++ + x);
                e[0] = rec0_of[f]; e[1] = ti; e[2] = tj;
              }
          }
        }
        break;
      }
      case 1:
        if (dev_maps) break;
        put(o_bdst, S.blk_dst.data(), S.blk_dst.size() * 4);
        put(o_rdst, S.b_dst.data(), S.b_dst.size() * 4);
        break;
      case 2: {
        put(o_fronts, S.fronts.data(), S.fronts.size() * sizeof(FrontDesc));
        FrontDesc* lv = reinterpret_cast<FrontDesc*>(h + o_fronts_lv);       // Gauss-Newton level order (the backward solve's index)
        for (size_t q = 0; q < S.fronts.size(); q++) {
          if (q < LF.size()) lv[q] = S.fronts[LF[q]];
          else memset(&lv[q], 0, sizeof(FrontDesc));             // fronts of the top block: not addressed through this table
        }
        put(o_children, S.children.data(), S.children.size() * 4);
        put(o_lf, S.level_fronts.data(), S.level_fronts.size() * 4);
        put(o_tf, S.top_fronts.data(), S.top_fronts.size() * 4);
        put(o_tc, S.top_children.data(), S.top_children.size() * 4);
        put(o_tb, S.top_blocks.data(), S.top_blocks.size() * 4);
        break;
      }
      case 3:
        put(o_rows, S.rows.data(), S.rows.size() * 4);
        if (!dev_maps) put(o_rel, S.rel.data(), S.rel.size() * 4);
        break;
      case 4:
        if (!dev_maps) put(o_inv, S.inv.data(), S.inv.size() * 4);
        if (early) break;
        put(o_asmp, S.asm_ptr.data(), S.asm_ptr.size() * 4);
        put(o_asms, S.asm_src.data(), S.asm_src.size() * 4);
        break;
      default:
        if (early) break;
        put(o_vperm, S.vperm.data(), S.vperm.size() * 4);
        put(o_ef, ef, (size_t)S.nE * 4);
        put(o_et, et, (size_t)S.nE * 4);
        put(o_orow, S.off_row.data(), S.off_row.size() * 4);
        put(o_ocol, S.off_col.data(), S.off_col.size() * 4);
        break;
    }
  });
  char* d = ctx->gn_arena.ptr;
  // a batch on the side stream still reads the structure this upload replaces
  rc = side_join_stream(ctx, ctx->stream);
  if (rc) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(d, h, blob_bytes, hipMemcpyHostToDevice, ctx->stream));
  D.fronts = (FrontDesc*)(d + o_fronts);
  D.fronts_lv = (FrontDesc*)(d + o_fronts_lv);
  D.rows = (int32_t*)(d + o_rows);
  D.children = (int32_t*)(d + o_children);
  D.rel = (int32_t*)(d + o_rel);
  D.inv = (int32_t*)(d + o_inv);
  D.blk_dst = (int32_t*)(d + o_bdst);
  D.b_dst = (int32_t*)(d + o_rdst);
  D.level_fronts = (int32_t*)(d + o_lf);
  D.tiles = (int32_t*)(d + o_tiles);
  D.work = (WorkRec*)(d + o_work);
  D.asm_ptr = (int32_t*)(d + o_asmp);
  D.asm_src = (int32_t*)(d + o_asms);
  D.vperm = (int32_t*)(d + o_vperm);
  D.ef = (int32_t*)(d + o_ef);
  D.et = (int32_t*)(d + o_et);
  D.off_row = (int32_t*)(d + o_orow);
  D.off_col = (int32_t*)(d + o_ocol);
  if (early) {
    const cgmr_ctx::StView& V = ctx->st_view;
    D.asm_ptr = V.asm_ptr; D.asm_src = V.asm_src; D.vperm = V.vperm; D.ef = V.ef; D.et = V.et; D.off_row = V.off_row; D.off_col = V.off_col;
  }
  D.cmask = (uint8_t*)(d + o_cmask);
  D.top_fronts = (int32_t*)(d + o_tf); D.top_children = (int32_t*)(d + o_tc); D.top_blocks = (int32_t*)(d + o_tb);
  D.top_nfronts = (int)S.top_fronts.size(); D.top_c0 = S.top_c0; D.top_ncols = 3 * S.top_nposes;
  D.top_nchild = (int)S.top_children.size(); D.top_nblk = (int)S.top_blocks.size() / 3;
  D.term = (double*)(d + o_term);
  D.Ablk = (double*)(d + o_A);
  D.bvec = (double*)(d + o_b);
  D.yvec = (double*)(d + o_y);
  D.xvec = (double*)(d + o_x);
  D.uvec = (double*)(d + o_u);
  D.Lbuf = (double*)(d + o_L);
  D.Ubuf = (double*)(d + o_U);
  if (dev_maps) {
    launch_build_maps(ctx->stream, D, ctx->st_view.offbase, (const int32_t*)(d + o_rec0));
    HIP_TRY(ctx, hipGetLastError());
  }
  D.Pan = (double*)(d + o_pan);
  D.pan_clean = false;
  D.pan_doubles = S.pan_doubles;
  D.chi2 = (double*)(d + o_chi);
  D.status = (int*)(d + o_status);
  D.ready = (int*)(d + o_ready);
  // the upper levels of the tree are solved backwards in one chained launch: as many levels as fit the workgroups that
  // are certainly resident together (the waits inside the launch cannot deadlock then); CGMR_BWD_CHAIN=0: one launch per level
  choose_bwd_chain(D, ctx->side_used ? 2 : 1, false);      // (the other half of the slots: the side stream's batches)
  choose_fwd_merge(D, ctx->side_used ? 2 : 1, false, ctx->fwd_merge_any);
  return 0;
}

// `n` extra copies of the numeric work space of the uploaded structure (the structure arrays are shared): independent
// numeric passes on the same graph -- one condensed graph per peer -- run concurrently on streams of their own.
int gn_replicas(cgmr_ctx* ctx, int n, std::vector<GnDevice>& out, size_t* stride_out) {
  const Symbolic& S = ctx->sym;
  const GnDevice& D0 = ctx->gn;
  BlobLayout N;
  size_t o_term = N.add<double>((size_t)34 * S.nE), o_A = N.add<double>((size_t)9 * (S.nf + S.nb)), o_b = N.add<double>((size_t)3 * S.nf),
         o_y = N.add<double>((size_t)3 * S.nf), o_x = N.add<double>((size_t)3 * S.nf), o_u = N.add<double>((size_t)3 * S.rows.size() + 3),
         o_L = N.add<double>((size_t)S.L_doubles + 1), o_U = N.add<double>((size_t)S.U_doubles + 1),
         o_pan = N.add<double>((size_t)S.pan_doubles + 2), o_chi = N.add<double>(8),
         o_status = N.add<int>(4), o_ready = N.add<int>(S.fronts.size() + 4), o_cmask = N.add<uint8_t>((size_t)S.nf + 16);
  const size_t per = (N.off + 255) & ~size_t(255);
  int rc = arena_reserve(ctx, ctx->rep_arena, per * (size_t)std::max(n, 1) + 256);
  if (rc) return rc;
  out.assign(n, D0);
  if (stride_out) *stride_out = per;
  for (int i = 0; i < n; i++) {
    char* d = ctx->rep_arena.ptr + per * (size_t)i;
    GnDevice& D = out[i];
    D.term = (double*)(d + o_term); D.Ablk = (double*)(d + o_A); D.bvec = (double*)(d + o_b); D.yvec = (double*)(d + o_y);
    D.xvec = (double*)(d + o_x); D.uvec = (double*)(d + o_u); D.Lbuf = (double*)(d + o_L); D.Ubuf = (double*)(d + o_U); D.Pan = (double*)(d + o_pan); D.pan_clean = false;
    D.chi2 = (double*)(d + o_chi); D.status = (int*)(d + o_status); D.ready = (int*)(d + o_ready); D.cmask = (uint8_t*)(d + o_cmask);
  }
  return 0;
}

// side streams for concurrent passes, created on first use
int aux_streams(cgmr_ctx* ctx, int n) {
  while ((int)ctx->aux.size() < n) {
    hipStream_t s = nullptr;
    hipEvent_t e = nullptr;
    HIP_TRY(ctx, hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    ctx->aux.push_back(s);
    ctx->aux_done.push_back(e);
  }
  if (!ctx->aux_fork) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->aux_fork, hipEventDisableTiming));
  return 0;
}

// The analysis of (nV, ef, et) into `sym`, which holds the analysis of (prev_nV, pef, pet) when have_prev: extended from it
// where the two lists allow that, from scratch otherwise (shared by prepare_structure and the host-only test hook).
int analyze_next(Symbolic& sym, bool have_prev, int prev_nV, const std::vector<int32_t>& pef, const std::vector<int32_t>& pet, int nV,
                 int nE, const int32_t* ef, const int32_t* et, const int32_t* hub_vertices, int n_hub_vertices, const AnalyzeHooks* hooks = nullptr) {
  int n_common = 0;
  if (have_prev && nV >= prev_nV && !pef.empty()) {
    const int lim = std::min(nE, (int)pef.size());
    while (n_common < lim && pef[n_common] == ef[n_common] && pet[n_common] == et[n_common]) n_common++;
  }
  const int old_nE = (int)pef.size();
  const bool grown = n_common > 0 && (n_common == old_nE || (n_hub_vertices > 0 && 2 * n_common >= old_nE));
  if (grown) {
    Symbolic old = std::move(sym);
    return analyze(nV, nullptr, nE, ef, et, sym, &old, n_common == old_nE && nE >= old_nE ? -1 : n_common, hub_vertices, n_hub_vertices, hooks);
  }
  return analyze(nV, nullptr, nE, ef, et, sym, nullptr, -1, hub_vertices, n_hub_vertices, hooks);
}

// Ordering + symbolic analysis + structure upload for the edge list (ef, et), or nothing at all when the context
// still holds them for exactly this list (g2o redoes buildStructure + cs_schol on every optimize() call,
// SURVEY.md 3.2; within one key frame -- optimize(1), covariance estimate, optimize(5), graph_slam.cpp:392-393,
// 315-320 -- and within one multi-robot round the list does not change).  The analysis does not look at the fixed
// flags (they are applied numerically, prepare_pass()), so a hit is bit-identical to a miss by construction.
int prepare_structure(cgmr_ctx* ctx, int nV, int nE, const int32_t* ef, const int32_t* et, int iters, const int32_t* hub_vertices,
                      int n_hub_vertices) {
  const bool hit = ctx->sym_cache_on && ctx->sym_valid && ctx->sym_nV == nV && (int)ctx->sym_ef.size() == nE &&
                   iters <= ctx->sym_chi_cap &&
                   (nE == 0 || (memcmp(ctx->sym_ef.data(), ef, sizeof(int32_t) * nE) == 0 &&
                                memcmp(ctx->sym_et.data(), et, sizeof(int32_t) * nE) == 0));
  if (hit) {
    ctx->sym_hits++;
    ctx->sym.t_order = ctx->sym.t_struct = ctx->sym.t_upload = 0;
    return 0;
  }
  // The key-frame pattern: the cached edge list plus vertices / edges appended at the end (graph_slam.cpp:197-267 adds a
  // vertex and a few edges per key frame; a multi-robot round adds a chunk).  The ordering is then extended instead of
  // recomputed (gn_symbolic.cpp: extend_order); everything downstream of the ordering is built as for a new graph.
  // A robot's list (hub_vertices given: cgmr_graph_optimize) is its own edges, appended to, followed by the edges received
  // from the peers, replaced every round: the two lists share their front only, the rest of the new one is checked against
  // the tree edge by edge (an edge to a hub -- the gauge of a received star -- always passes).
  const bool have_prev = ctx->sym_cache_on && ctx->sym_valid;
  ctx->sym_valid = false;
  static const bool asm_device = !(getenv("CGMR_ASM_DEVICE") && atoi(getenv("CGMR_ASM_DEVICE")) == 0);
  AnalyzeHooks hooks;
  int hook_rc = 0;
  static const bool maps_device = !(getenv("CGMR_MAPS_DEVICE") && atoi(getenv("CGMR_MAPS_DEVICE")) == 0);
  hooks.maps_on_device = asm_device && maps_device;
  if (asm_device)
    hooks.blocks_ready = [&](const Symbolic& S, const int32_t* offbase) { hook_rc = gn_upload_early(ctx, S, ef, et, offbase); return hook_rc ? -100 : 0; };
  int rc = analyze_next(ctx->sym, have_prev, ctx->sym_nV, ctx->sym_ef, ctx->sym_et, nV, nE, ef, et, hub_vertices, n_hub_vertices, &hooks);
  if (rc == -100) return hook_rc;                                   // (the device pass of the analysis failed: its error is set)
  if (rc == 0 && ctx->sym.extended) ctx->sym_extended++; else ctx->sym_misses++;
  if (rc) return set_err(ctx, CGMR_E_INVALID, "graph structure rejected (edge index out of range)");
  const int chi_cap = std::max(iters, 30);
  const double tu0 = wall_s();
  rc = gn_upload(ctx, ctx->sym, ef, et, chi_cap);
  if (rc) return rc;
  ctx->sym.t_upload = wall_s() - tu0;
  if (ctx->sym_cache_on) {
    ctx->sym_nV = nV;
    ctx->sym_ef.assign(ef, ef + nE);
    ctx->sym_et.assign(et, et + nE);
    ctx->sym_valid = true;
  }
  ctx->sym_chi_cap = chi_cap;
  return 0;
}

// Per numeric pass: the column mask (fixed vertices; vertices whose edges are all switched off when only the
// first n_active edges take part), the status words.  ctx->vmask keeps the per-vertex flags for the caller.
int prepare_pass(cgmr_ctx* ctx, const uint8_t* fixed, int nE, const int32_t* ef, const int32_t* et, int n_active, int slot,
                 int nslots) {
  return prepare_pass_on(ctx, ctx->gn, ctx->stream, fixed, nE, ef, et, n_active, slot, nslots);
}

int prepare_pass_on(cgmr_ctx* ctx, GnDevice& D, hipStream_t st, const uint8_t* fixed, int nE, const int32_t* ef, const int32_t* et,
                    int n_active, int slot, int nslots, bool upload, char* stage) {
  const Symbolic& S = ctx->sym;
  ctx->vmask.assign(S.nV, 0);
  if (fixed) for (int v = 0; v < S.nV; v++) ctx->vmask[v] = fixed[v] ? 1 : 0;
  if (n_active < nE) {
    std::vector<uint8_t> live(S.nV, 0);
    for (int k = 0; k < n_active; k++) { live[ef[k]] = 1; live[et[k]] = 1; }
    for (int v = 0; v < S.nV; v++) if (!live[v]) ctx->vmask[v] = 1;
  }
  if (upload) HIP_TRY(ctx, hipMemsetAsync(D.status, 0, 16, st));
  if (S.nf == 0) return 0;
  // several passes may be queued without a host synchronisation in between (one condensed graph per peer): each
  // stages its mask in a slot of its own
  if (!stage) {
    int rc = pinned_mask_reserve(ctx, (size_t)S.nf * std::max(nslots, 1));
    if (rc) return rc;
    stage = ctx->pinned_mask;
  }
  char* pm = stage + (size_t)S.nf * slot;
  for (int c = 0; c < S.nf; c++) pm[c] = (char)ctx->vmask[S.perm[c]];
  if (upload) HIP_TRY(ctx, hipMemcpyAsync(D.cmask, pm, (size_t)S.nf, hipMemcpyHostToDevice, st));
  return 0;
}

// the passes of a batch (GnDevice::njobs) start from the masks staged in slots 0 .. njobs-1 and from clean status words
int prepare_batch_on(cgmr_ctx* ctx, GnDevice& D, hipStream_t st) {
  const Symbolic& S = ctx->sym;
  HIP_TRY(ctx, hipMemset2DAsync(D.status, (size_t)D.job_stride, 0, 16, (size_t)D.njobs, st));
  if (S.nf == 0) return 0;
  HIP_TRY(ctx, hipMemcpy2DAsync(D.cmask, (size_t)D.job_stride, ctx->pinned_mask, (size_t)S.nf, (size_t)S.nf, (size_t)D.njobs, hipMemcpyHostToDevice, st));
  return 0;
}

// profiling mode: add up the event pairs of the launches since the last collection (call after a stream sync)
void profile_collect(cgmr_ctx* ctx) {
  for (size_t k = 0; k < ctx->ev_cls.size(); k++) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, ctx->ev_pool[2 * k], ctx->ev_pool[2 * k + 1]) == hipSuccess) ctx->ksec[ctx->ev_cls[k]] += 1e-3 * ms;
  }
  ctx->ev_cls.clear();
}

struct KTimer {   // optional per-launch-class timing (profiling mode only)
  cgmr_ctx* ctx;
  hipStream_t st;
  template <typename Fn>
  void run(int cls, int nlaunch, Fn&& fn) {
    static const bool trace = getenv("CGMR_TRACE_LAUNCHES") != nullptr;      // debugging aid: name every launch, sync after it
    if (trace) {
      fprintf(stderr, "[cgmr] launch class %d (0 lin 1 asm 2 chi2 3 factor 4 update 5 top 6 bwd 7 poses 8 factor+update)\n", cls);
      fn();
      hipError_t e = hipStreamSynchronize(st);
      fprintf(stderr, "[cgmr]   done: %s\n", hipGetErrorString(e));
      return;
    }
    if (!ctx->profiling || st != ctx->stream || 2 * (ctx->ev_cls.size() + 1) > ctx->ev_pool.size()) { fn(); return; }
    const size_t k = ctx->ev_cls.size();
    (void)hipEventRecord(ctx->ev_pool[2 * k], ctx->stream);
    fn();
    (void)hipEventRecord(ctx->ev_pool[2 * k + 1], ctx->stream);
    ctx->ev_cls.push_back(cls);
    ctx->klaunch[cls] += nlaunch;
  }
};

// one Gauss-Newton pass on the uploaded structure: linearise + chi2 [+ assemble + factor [+ solve + update]]
void gn_pass(cgmr_ctx* ctx, double* d_poses, const GnEdges& Ed, int it, bool chi_only,
             bool solve_and_update, bool write_l11c) {
  gn_pass_on(ctx, ctx->gn, ctx->stream, d_poses, Ed, it, chi_only, solve_and_update, write_l11c);
}

// the same on an explicit device view (the context's, or a replica with its own numeric work space) and stream
void gn_pass_on(cgmr_ctx* ctx, GnDevice& D, hipStream_t st, double* d_poses, const GnEdges& Ed, int it, bool chi_only,
                bool solve_and_update, bool write_l11c) {
  KTimer T{ctx, st};
  T.run(0, 1, [&] { launch_linearize(st, D, d_poses, Ed, chi_only ? 1 : 0); });
  if (chi_only || D.nf == 0) {
    T.run(2, 1, [&] { launch_chi2(st, D, D.chi2 + it); });
    return;
  }
  // the assembled panels start from zero: H blocks and b (k_assemble), then the children's contributions level by level
  // (zeroing them for the next pass on a side stream underneath this pass's backward solve was measured slower: 5.79
  // instead of 5.47 ms per optimize(10) -- the 36 MB of writes slow the chained solve's hops more than the 8 us they hide)
  // ... but the top-block launch, one workgroup on an idle chip, clears them for the next pass with its other workgroups
  if (D.pan_doubles > 0 && !D.pan_clean) {
    if (D.njobs > 1) (void)hipMemset2DAsync(D.Pan, (size_t)D.job_stride, 0, sizeof(double) * (size_t)D.pan_doubles, (size_t)D.njobs, st);
    else (void)hipMemsetAsync(D.Pan, 0, sizeof(double) * (size_t)D.pan_doubles, st);
  }
  D.pan_clean = false;
  T.run(1, 1, [&] { launch_assemble(st, D); });                // + the chi2 sum of this iteration (slot = iterations done)
  static const bool trace = getenv("CGMR_TRACE_LAUNCHES") != nullptr;
  if (trace)
    fprintf(stderr, "[cgmr] arena %p .. %p; work %p rel %p Pan %p Ablk %p bvec %p yvec %p uvec %p Lbuf %p Ubuf %p chi2 %p\n",
            (void*)ctx->gn_arena.ptr, (void*)(ctx->gn_arena.ptr + ctx->gn_arena.cap), (void*)D.work, (void*)D.rel, (void*)D.Pan,
            (void*)D.Ablk, (void*)D.bvec, (void*)D.yvec, (void*)D.uvec, (void*)D.Lbuf, (void*)D.Ubuf, (void*)D.chi2);
  for (int l = 0; l < D.nlevels; l++) {
    if (trace) {
      int maxr = 0, maxc = 0;
      for (int q = D.h_level_ptr[l]; q < D.h_level_ptr[l + 1]; q++) {
        const FrontDesc& F = ctx->sym.fronts[ctx->sym.gn_level_fronts[q]];
        maxr = std::max(maxr, 3 * F.ns);
        for (int k = 0; k < F.nchild; k++) maxc = std::max(maxc, 3 * ctx->sym.fronts[ctx->sym.children[F.child_off + k]].ns);
      }
      if (D.h_level_ptr[l + 1] - D.h_level_ptr[l] <= 2)
        for (int q = D.h_level_ptr[l]; q < D.h_level_ptr[l + 1]; q++) {
          const FrontDesc& F = ctx->sym.fronts[ctx->sym.gn_level_fronts[q]];
          fprintf(stderr, "[cgmr]   front nc %d ns %d na %d nchild %d a_cnt %d L_off %lld U_off %lld:", F.nc, F.ns, F.na, F.nchild, F.a_cnt, (long long)F.L_off, (long long)F.U_off);
          for (int k = 0; k < F.nchild; k++) { const FrontDesc& G = ctx->sym.fronts[ctx->sym.children[F.child_off + k]]; fprintf(stderr, " child(ns %d na %d U_off %lld)", G.ns, G.na, (long long)G.U_off); }
          fprintf(stderr, "\n");
        }
      fprintf(stderr, "[cgmr] level %d: %d fronts, %d work items, max r %d, max child rows %d\n", l,
              D.h_level_ptr[l + 1] - D.h_level_ptr[l], D.h_work_ptr[l + 1] - D.h_work_ptr[l], maxr, maxc);
    }
    if (l < (int)D.h_level_merge.size() && D.h_level_merge[l]) { T.run(8, 1, [&] { launch_front_level(st, D, l, write_l11c); }); continue; }
    T.run(3, 1, [&] { launch_factor_level(st, D, l, write_l11c); });
    if (D.h_tile_ptr[l + 1] > D.h_tile_ptr[l]) T.run(4, 1, [&] { launch_update_level(st, D, l); });
  }
  // the top of the tree in one launch: assembly, factorisation, forward and backward solve of the block's columns
  static const bool clear_in_top = !(getenv("CGMR_CLEAR_IN_TOP") && atoi(getenv("CGMR_CLEAR_IN_TOP")) == 0);
  // (the chained backward solve works with L11^-1 of every front: made by the idle workgroups of the top-block launch)
  const bool chain = solve_and_update && D.bwd_chain_level < D.nlevels;
  if (D.top_nfronts > 0) {
    T.run(5, 1, [&] { launch_top_block(st, D, /*store_l=*/write_l11c, write_l11c, clear_in_top, chain); });
    D.pan_clean = clear_in_top && D.pan_doubles > 0;
  } else if (chain) {
    T.run(5, 1, [&] { launch_invert_fronts(st, D); });
  }
  if (!solve_and_update) return;
  // (the forward solve L y = b rides through k_front_factor as an extra row of every front)
  if (D.bwd_chain_level < D.nlevels) T.run(6, 1, [&] { launch_bwd_chain(st, D); });
  for (int l = D.bwd_chain_level - 1; l >= 0; l--) T.run(6, 1, [&] { launch_bwd_level(st, D, l); });
  T.run(7, 1, [&] { launch_update(st, D, d_poses); });
}

int gn_run(cgmr_ctx* ctx, int nV, double* d_poses, const uint8_t* fixed, int nE, const int32_t* ef,
           const int32_t* et, const GnEdges& Ed, int iters, double* chi2_out, const int32_t* hub_vertices, int n_hub_vertices) {
  double t0 = wall_s();
  Symbolic& S = ctx->sym;
  int rc = prepare_structure(ctx, nV, nE, ef, et, iters, hub_vertices, n_hub_vertices);
  if (rc) return rc;
  double t1 = wall_s();
  rc = prepare_pass(ctx, fixed, nE, ef, et, Ed.n_active, 0, 1);
  if (rc) return rc;
  double t2 = wall_s();
  GnDevice& D = ctx->gn;
  hipStream_t st = ctx->stream;
  HIP_TRY(ctx, hipEventRecord(ctx->ev0, st));
  // Every launch of a GN iteration is the same whatever the iteration's number (status[1] on the device supplies the
  // chi2 slot and the failure tag): CGMR_GRAPH=1 captures one iteration into a hipGraph and replays it.
  static const bool graph_mode = getenv("CGMR_GRAPH") && atoi(getenv("CGMR_GRAPH")) != 0;
  static const bool trace_launches = getenv("CGMR_TRACE_LAUNCHES") != nullptr;
  hipGraph_t graph = nullptr;
  hipGraphExec_t graph_exec = nullptr;
  if (graph_mode && !ctx->profiling && !trace_launches && iters >= 2 && st != nullptr && D.nf > 0) {
    gn_init_kernels();
    HIP_TRY(ctx, hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    gn_pass(ctx, d_poses, Ed, 0, false, true, false);
    HIP_TRY(ctx, hipStreamEndCapture(st, &graph));
    HIP_TRY(ctx, hipGraphInstantiate(&graph_exec, graph, nullptr, nullptr, 0));
    for (int it = 0; it < iters; it++) HIP_TRY(ctx, hipGraphLaunch(graph_exec, st));
    gn_pass(ctx, d_poses, Ed, iters, true, true, false);
  } else {
    for (int it = 0; it <= iters; it++) gn_pass(ctx, d_poses, Ed, it, it == iters, true, false);
  }
  HIP_TRY(ctx, hipEventRecord(ctx->ev1, st));
  const double t3 = wall_s();
  // read back chi2 + status
  std::vector<double> chi(iters + 1);
  int status4[4] = {0, 0, 0, 0};
  HIP_TRY(ctx, hipMemcpyAsync(chi.data(), D.chi2, sizeof(double) * (iters + 1), hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipMemcpyAsync(status4, D.status, sizeof status4, hipMemcpyDeviceToHost, st));
  // the caller's host copy of the estimates, in the same wait (a robot graph whose peers have asked for condensed graphs
  // picks their gauges from it right after the solve)
  double* const poses_host = ctx->poses_out_host;
  ctx->poses_out_host = nullptr;
  if (poses_host && nV > 0) HIP_TRY(ctx, hipMemcpyAsync(poses_host, d_poses, 24 * (size_t)nV, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  HIP_TRY(ctx, hipGetLastError());
  {
    // CGMR_GN_TRACE: where a solve's wall time goes beside the analysis -- queueing the launches, waiting for the stream
    static const bool gn_trace = getenv("CGMR_GN_TRACE") != nullptr;
    if (gn_trace) {
      const double t4 = wall_s();
      ctx->trace_n++; ctx->trace_sum[0] += t1 - t0; ctx->trace_sum[1] += t2 - t1; ctx->trace_sum[2] += t3 - t2; ctx->trace_sum[3] += t4 - t3;
    }
  }
  if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
  if (graph) (void)hipGraphDestroy(graph);
  if (status4[2] != 0 && status4[0] > 0) {
    // A bounded wait of the chained backward solve ran out (status[2]): a hand-off between workgroups that never arrived,
    // not a numerical failure.  The iteration it happened in (status[0] - 1) and the later ones were not applied
    // (k_update_poses leaves the poses alone once status[0] is set), so they are repeated from the poses as they stand with
    // one backward launch per tree level -- no in-kernel waits -- and the call goes on as if nothing had happened.
    ctx->gn_timeouts++;
    ctx->fwd_merge_any = false;                                  // (the next structures merge only what is certainly resident)
    const int it0 = status4[0] - 1;
    const int chain_was = D.bwd_chain_level;
    D.bwd_chain_level = D.nlevels;
    const std::vector<uint8_t> merge_was = D.h_level_merge;
    D.h_level_merge.assign(D.nlevels, 0);
    const int fresh[4] = {0, it0, 0, 0};
    HIP_TRY(ctx, hipMemcpyAsync(D.status, fresh, sizeof fresh, hipMemcpyHostToDevice, st));
    for (int it = it0; it <= iters; it++) gn_pass(ctx, d_poses, Ed, it, it == iters, true, false);
    HIP_TRY(ctx, hipMemcpyAsync(chi.data(), D.chi2, sizeof(double) * (iters + 1), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(status4, D.status, sizeof status4, hipMemcpyDeviceToHost, st));
    if (poses_host && nV > 0) HIP_TRY(ctx, hipMemcpyAsync(poses_host, d_poses, 24 * (size_t)nV, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    HIP_TRY(ctx, hipGetLastError());
    D.bwd_chain_level = chain_was;
    D.h_level_merge = merge_was;
    if (status4[2] != 0) return set_err(ctx, CGMR_E_TIMEOUT, "backward solve: a bounded device-side wait ran out twice");
  }
  const int status = status4[0];
  if (ctx->profiling) profile_collect(ctx);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
  if (chi2_out) memcpy(chi2_out, chi.data(), sizeof(double) * (iters + 1));
  ctx->timing[0] = S.t_order;
  ctx->timing[1] = S.t_struct;
  ctx->timing[2] = (t2 - t1) + S.t_upload;     // structure blob (host staging + H2D enqueue) + per-pass masks
  ctx->timing[3] = 1e-3 * ms;
  ctx->timing[4] = wall_s() - t0;
  if (status != 0)
    return set_err(ctx, CGMR_E_CHOLESKY_BASE - (status - 1),
                   "Cholesky failed (non-positive pivot) in GN iteration %d; poses left at the last good update",
                   status - 1);
  return CGMR_OK;
}

// SparseOptimizer::computeInitialGuess with unit edge cost [g2o-recalled]: breadth-first from the fixed
// vertices over the given edges (edge order breaks ties), x_to = x_from * z or x_from = x_to * z^-1.
void initial_guess_host(int nV, double* poses, const uint8_t* fixed, int nE, const int32_t* ef, const int32_t* et,
                        const double* meas) {
  auto norm = [](double t) {
    const double pi = 3.14159265358979323846;
    if (t >= -pi && t < pi) return t;
    return t - 2 * pi * std::floor((t + pi) / (2 * pi));
  };
  std::vector<int> deg(nV + 1, 0);
  for (int k = 0; k < nE; k++) { deg[ef[k] + 1]++; deg[et[k] + 1]++; }
  for (int v = 0; v < nV; v++) deg[v + 1] += deg[v];
  std::vector<int> inc(2 * (size_t)nE + 1), pos(deg.begin(), deg.end() - 1);
  for (int k = 0; k < nE; k++) { inc[pos[ef[k]]++] = k; inc[pos[et[k]]++] = k; }
  std::vector<uint8_t> seen(nV, 0);
  std::vector<int> queue;
  queue.reserve(nV);
  for (int v = 0; v < nV; v++) if (fixed[v] && deg[v + 1] > deg[v]) { seen[v] = 1; queue.push_back(v); }
  for (size_t qh = 0; qh < queue.size(); qh++) {
    int u = queue[qh];
    for (int p = deg[u]; p < deg[u + 1]; p++) {
      int k = inc[p];
      int w = (ef[k] == u) ? et[k] : ef[k];
      if (seen[w]) continue;
      seen[w] = 1;
      const double* a = poses + 3 * (size_t)u;
      double z[3] = {meas[3 * k], meas[3 * k + 1], meas[3 * k + 2]};
      if (ef[k] != u) {   // z^-1
        double c = std::cos(z[2]), s = std::sin(z[2]);
        double ix = -(c * z[0] + s * z[1]), iy = -(-s * z[0] + c * z[1]);
        z[0] = ix; z[1] = iy; z[2] = -z[2];
      }
      double c = std::cos(a[2]), s = std::sin(a[2]);
      double* o = poses + 3 * (size_t)w;
      o[0] = a[0] + c * z[0] - s * z[1];
      o[1] = a[1] + s * z[0] + c * z[1];
      o[2] = norm(a[2] + z[2]);
      queue.push_back(w);
    }
  }
}

// Shared driver of cgmr_marginals / cgmr_covariance_estimate / cgmr_condense (host pointers).
//   mode 0: marginals at `poses` with `fixed`;  mode 1: covariance estimate (gauge);  mode 2: condense (gauge)
int marginal_driver(cgmr_ctx* ctx, int mode, int nV, const double* poses, const uint8_t* fixed_in, int nE,
                    const int32_t* ef, const int32_t* et, const double* meas, const double* info, int gauge, int nK,
                    const int32_t* query, int32_t* to_out, double* est_out, double* info_out, double* cov_out) {
  if (nV <= 0 || nE < 0 || nK < 0 || !poses || (nE > 0 && (!ef || !et || !meas || !info)) || (nK > 0 && !query))
    return set_err(ctx, CGMR_E_INVALID, "marginals: null or negative argument");
  if (mode != 0 && (gauge < 0 || gauge >= nV)) return set_err(ctx, CGMR_E_INVALID, "gauge index out of range");
  for (int k = 0; k < nK; k++)
    if (query[k] < 0 || query[k] >= nV) return set_err(ctx, CGMR_E_INVALID, "query index out of range");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  std::vector<uint8_t> fixed(nV, 0);
  std::vector<double> work(poses, poses + 3 * (size_t)nV);             // pushState: the caller's poses stay untouched
  if (mode == 0) { if (!fixed_in) return set_err(ctx, CGMR_E_INVALID, "fixed flags missing"); fixed.assign(fixed_in, fixed_in + nV); }
  else {
    fixed[gauge] = 1;                                                   // fixGauge: every other vertex is freed
    initial_guess_host(nV, work.data(), fixed.data(), nE, ef, et, meas);
  }
  // query list (condense: everything but the gauge)
  std::vector<int32_t> q;
  for (int k = 0; k < nK; k++) if (mode != 2 || query[k] != gauge) q.push_back(query[k]);
  const int nq = (int)q.size();
  if (cov_out && mode != 2) memset(cov_out, 0, sizeof(double) * 9 * (size_t)nK);
  Symbolic& S = ctx->sym;
  int rc = prepare_structure(ctx, nV, nE, ef, et, 1);
  if (rc) return rc;
  rc = prepare_pass(ctx, fixed.data(), nE, ef, et, nE, 0, 1);
  if (rc) return rc;
  GnDevice& D = ctx->gn;
  hipStream_t st = ctx->stream;
  if (D.nf == 0 || nq == 0) { HIP_TRY(ctx, hipStreamSynchronize(st)); return mode == 2 ? 0 : CGMR_OK; }
  const int m = ((4 * nq + 15) / 16) * 16;          // 4 columns of Y per query (3 + 1 padding): a query never straddles a 16-column tile
  const int n = 3 * D.nf;
  const int chunk = 2048, nchunk = (n + chunk - 1) / chunk;
  // staging: poses | meas | info | qcol | qvert | Y | Uv | part | G | cov | est | info_out | flags
  struct L2 { size_t off = 0; size_t add(size_t b) { off = (off + 255) & ~size_t(255); size_t o = off; off += b; return o; } } L;
  size_t o_p = L.add(24 * (size_t)nV), o_m = L.add(24 * (size_t)nE), o_i = L.add(48 * (size_t)nE), o_qc = L.add(4 * (size_t)nq),
         o_qv = L.add(4 * (size_t)nq), o_Y = L.add(8 * (size_t)n * m), o_U = L.add(8 * ((size_t)3 * S.rows.size() + 3) * m),
         o_part = L.add(8 * (size_t)nchunk * 16 * m), o_G = L.add(8 * (size_t)16 * m), o_cov = L.add(72 * (size_t)nq),
         o_est = L.add(24 * (size_t)nq), o_io = L.add(48 * (size_t)nq), o_fl = L.add(4 * (size_t)nq),
         o_live = L.add((size_t)std::max(ctx->gn.nfronts, 1) * (m / 16));
  rc = arena_reserve(ctx, ctx->io_arena, L.off + 256);
  if (rc) return rc;
  char* d = ctx->io_arena.ptr;
  std::vector<int32_t> qcol(nq);
  for (int k = 0; k < nq; k++) qcol[k] = ctx->vmask[q[k]] ? -1 : S.vperm[q[k]];     // fixed / inactive: zeros
  HIP_TRY(ctx, hipMemcpyAsync(d + o_p, work.data(), 24 * (size_t)nV, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(d + o_m, meas, 24 * (size_t)nE, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(d + o_i, info, 48 * (size_t)nE, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(d + o_qc, qcol.data(), 4 * (size_t)nq, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(d + o_qv, q.data(), 4 * (size_t)nq, hipMemcpyHostToDevice, st));
  double* dp = (double*)(d + o_p);
  // the Hessian of this iteration (linearised at the initial guess) is what computeMarginals sees [g2o-recalled];
  // for the condensed graph the iteration is completed first: the factor stays valid, the poses move on
  GnEdges Ed;
  Ed.meas_a = (const double*)(d + o_m); Ed.info_a = (const double*)(d + o_i); Ed.nA = nE; Ed.n_active = nE;
  gn_pass(ctx, dp, Ed, 0, false, mode == 2, /*write_l11c=*/true);
  launch_marginals(st, D, nq, (const int32_t*)(d + o_qc), m, (double*)(d + o_Y), (double*)(d + o_U), (double*)(d + o_part),
                   (double*)(d + o_G), (double*)(d + o_cov), chunk, nchunk, (uint8_t*)(d + o_live));
  if (mode == 2)
    launch_label(st, nq, (const int32_t*)(d + o_qv), gauge, dp, (const double*)(d + o_cov), (double*)(d + o_est),
                 (double*)(d + o_io), (int*)(d + o_fl));
  int status4[4] = {0, 0, 0, 0};
  std::vector<double> cov(9 * (size_t)nq);
  HIP_TRY(ctx, hipMemcpyAsync(cov.data(), d + o_cov, 72 * (size_t)nq, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipMemcpyAsync(status4, D.status, sizeof status4, hipMemcpyDeviceToHost, st));
  if (mode == 2) {
    HIP_TRY(ctx, hipMemcpyAsync(est_out, d + o_est, 24 * (size_t)nq, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(info_out, d + o_io, 48 * (size_t)nq, hipMemcpyDeviceToHost, st));
  }
  HIP_TRY(ctx, hipStreamSynchronize(st));
  HIP_TRY(ctx, hipGetLastError());
  if (status4[2] != 0) { ctx->gn_timeouts++; ctx->fwd_merge_any = false; return set_err(ctx, CGMR_E_TIMEOUT, "backward solve: a bounded device-side wait ran out"); }
  if (status4[0] != 0) return set_err(ctx, CGMR_E_CHOLESKY_BASE, "Cholesky failed while computing marginals");
  if (mode == 2) {
    for (int k = 0; k < nq; k++) to_out[k] = q[k];
    if (cov_out) memcpy(cov_out, cov.data(), 72 * (size_t)nq);
    return nq;
  }
  // scatter back in query order; fixed / inactive queries keep zeros
  for (int k = 0; k < nq; k++)
    if (qcol[k] >= 0) memcpy(cov_out + 9 * (size_t)k, cov.data() + 9 * (size_t)k, 72);
  return CGMR_OK;
}

}  // namespace cgmr

using namespace cgmr;

extern "C" {

int cgmr_version(void) { return 102; }   // 102 (round 5): cgmr_match_last_redo_pairs / _path_counts, cgmr_graph_failed_batches, cgmr_comm_info added; nothing changed or removed

int cgmr_ctx_create(int device, void* hip_stream, cgmr_ctx** out) {
  if (!out) return CGMR_E_INVALID;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return CGMR_E_NO_DEVICE;
  if (hipSetDevice(device) != hipSuccess) return CGMR_E_NO_DEVICE;
  cgmr_ctx* ctx = new cgmr_ctx();
  ctx->device = device;
  ctx->fwd_merge_any = !(getenv("CGMR_FWD_MERGE_ANY") && atoi(getenv("CGMR_FWD_MERGE_ANY")) == 0);
  if (hip_stream) { ctx->stream = (hipStream_t)hip_stream; ctx->own_stream = false; }
  else {
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return CGMR_E_HIP; }
    ctx->own_stream = true;
  }
  if (hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess ||
      hipEventCreate(&ctx->ev_a) != hipSuccess || hipEventCreate(&ctx->ev_b) != hipSuccess) {
    cgmr_ctx_destroy(ctx);
    return CGMR_E_HIP;
  }
  *out = ctx;
  return CGMR_OK;
}

void cgmr_ctx_destroy(cgmr_ctx* ctx) {
  if (!ctx) return;
  // teardown: nothing useful can be done with a failing free, the statuses are dropped on purpose -- and a HIP runtime that
  // is already shutting down (a binding's garbage collector at interpreter exit) throws instead of returning an error
  try {
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->trace_n > 0)
    fprintf(stderr, "[gn] %lld solves: analysis + upload %.3f ms, masks %.3f ms, queueing the launches %.3f ms, waiting %.3f ms per solve\n", (long long)ctx->trace_n,
            1e3 * ctx->trace_sum[0] / ctx->trace_n, 1e3 * ctx->trace_sum[1] / ctx->trace_n, 1e3 * ctx->trace_sum[2] / ctx->trace_n, 1e3 * ctx->trace_sum[3] / ctx->trace_n);
  for (auto& q : ctx->graveyard) (void)hipFree(q.first);
  if (ctx->gn_arena.ptr) (void)hipFree(ctx->gn_arena.ptr);
  if (ctx->io_arena.ptr) (void)hipFree(ctx->io_arena.ptr);
  if (ctx->mt_arena.ptr) (void)hipFree(ctx->mt_arena.ptr);
  if (ctx->mtab_arena.ptr) (void)hipFree(ctx->mtab_arena.ptr);
  if (ctx->rep_arena.ptr) (void)hipFree(ctx->rep_arena.ptr);
  if (ctx->mg_arena.ptr) (void)hipFree(ctx->mg_arena.ptr);
  if (ctx->st_arena.ptr) (void)hipFree(ctx->st_arena.ptr);
  if (ctx->pinned_st) (void)hipHostFree(ctx->pinned_st);
  if (ctx->ev_st_copied) (void)hipEventDestroy(ctx->ev_st_copied);
  for (hipStream_t a : ctx->aux) { (void)hipStreamSynchronize(a); (void)hipStreamDestroy(a); }
  if (ctx->side) { (void)hipStreamSynchronize(ctx->side); (void)hipStreamDestroy(ctx->side); }
  for (hipEvent_t e : {ctx->side_fork, ctx->side_tail}) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : ctx->aux_done) (void)hipEventDestroy(e);
  if (ctx->aux_fork) (void)hipEventDestroy(ctx->aux_fork);
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  if (ctx->pinned_mask) (void)hipHostFree(ctx->pinned_mask);
  for (hipEvent_t e : {ctx->ev0, ctx->ev1, ctx->ev_a, ctx->ev_b})
    if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  } catch (...) {
  }
  delete ctx;
}

const char* cgmr_last_error(const cgmr_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

void* cgmr_ctx_stream(const cgmr_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int cgmr_ctx_synchronize(cgmr_ctx* ctx) {
  if (!ctx) return CGMR_E_INVALID;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return CGMR_OK;
}

int cgmr_gn_optimize_dev(cgmr_ctx* ctx, int nV, double* d_poses, const uint8_t* fixed, int nE,
                         const int32_t* from_idx, const int32_t* to_idx, const double* d_meas,
                         const double* d_info, int iters, double* chi2_out) {
  if (!ctx) return CGMR_E_INVALID;
  if (nV < 0 || nE < 0 || iters < 0 || (nV > 0 && (!d_poses || !fixed)) ||
      (nE > 0 && (!from_idx || !to_idx || !d_meas || !d_info)))
    return set_err(ctx, CGMR_E_INVALID, "cgmr_gn_optimize: null or negative argument");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  GnEdges Ed;
  Ed.meas_a = d_meas; Ed.info_a = d_info; Ed.nA = nE; Ed.n_active = nE;
  return gn_run(ctx, nV, d_poses, fixed, nE, from_idx, to_idx, Ed, iters, chi2_out);
}

int cgmr_gn_optimize(cgmr_ctx* ctx, int nV, double* poses, const uint8_t* fixed, int nE, const int32_t* from_idx,
                     const int32_t* to_idx, const double* meas, const double* info, int iters, double* chi2_out) {
  if (!ctx) return CGMR_E_INVALID;
  if (nV < 0 || nE < 0 || iters < 0 || (nV > 0 && (!poses || !fixed)) ||
      (nE > 0 && (!from_idx || !to_idx || !meas || !info)))
    return set_err(ctx, CGMR_E_INVALID, "cgmr_gn_optimize: null or negative argument");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  size_t bp = sizeof(double) * 3 * (size_t)nV, bm = sizeof(double) * 3 * (size_t)nE, bi = sizeof(double) * 6 * (size_t)nE;
  size_t op = 0, om = (bp + 255) & ~size_t(255), oi = (om + bm + 255) & ~size_t(255);
  int rc = arena_reserve(ctx, ctx->io_arena, oi + bi + 256);
  if (rc) return rc;
  char* d = ctx->io_arena.ptr;
  HIP_TRY(ctx, hipMemcpyAsync(d + op, poses, bp, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d + om, meas, bm, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d + oi, info, bi, hipMemcpyHostToDevice, ctx->stream));
  GnEdges Ed;
  Ed.meas_a = (const double*)(d + om); Ed.info_a = (const double*)(d + oi); Ed.nA = nE; Ed.n_active = nE;
  rc = gn_run(ctx, nV, (double*)(d + op), fixed, nE, from_idx, to_idx, Ed, iters, chi2_out);
  if (rc == CGMR_OK || rc <= CGMR_E_CHOLESKY_BASE) {
    hipError_t e = hipMemcpyAsync(poses, d + op, bp, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return set_err(ctx, CGMR_E_HIP, "pose read-back: %s", hipGetErrorString(e));
  }
  return rc;
}

static void symbolic_info_out(const Symbolic& S, int nV, int64_t out[16], int32_t* perm_out) {
  out[0] = S.nf; out[1] = S.nb; out[2] = (int64_t)S.fronts.size(); out[3] = (int64_t)S.level_ptr.size() - 1;
  out[4] = S.L_doubles; out[5] = S.U_doubles; out[6] = S.max_ns; out[7] = (int64_t)S.flops;
  out[8] = (int64_t)(1e6 * S.t_order); out[9] = (int64_t)(1e6 * S.t_struct);
  out[10] = out[11] = out[12] = 0;
  out[13] = (int64_t)S.gn_level_ptr.size() - 1; out[14] = (int64_t)S.top_fronts.size(); out[15] = 3 * (int64_t)S.top_nposes;
  for (const FrontDesc& F : S.fronts) {
    out[12] += pan_size(F.ns) * F.pan_slots;
    out[10] = std::max<int64_t>(out[10], F.nchild);
    if (F.ns >= 1 && F.ns <= 32) out[11] = std::max<int64_t>(out[11], F.nchild);
  }
  if (perm_out) memcpy(perm_out, S.vperm.data(), sizeof(int32_t) * nV);
}

int cgmr_gn_symbolic_info(int nV, const uint8_t* fixed, int nE, const int32_t* from_idx, const int32_t* to_idx,
                          int64_t out[16], int32_t* perm_out) {
  if (nV < 0 || nE < 0 || !out) return CGMR_E_INVALID;
  Symbolic S;
  (void)fixed;          // the solver applies the fixed flags numerically: they are not part of the analysis
  int rc = analyze(nV, nullptr, nE, from_idx, to_idx, S);
  if (rc) return CGMR_E_INVALID;
  symbolic_info_out(S, nV, out, perm_out);
  return CGMR_OK;
}

int cgmr_gn_symbolic_info_grown(int nV0, int nE0, int n_steps, const int32_t* nV_step, const int32_t* nE_step,
                                const int32_t* from_idx, const int32_t* to_idx, int64_t out[16], int32_t* perm_out,
                                int32_t* n_extended_out) {
  if (nV0 < 0 || nE0 < 0 || n_steps < 0 || !out || (n_steps > 0 && (!nV_step || !nE_step))) return CGMR_E_INVALID;
  Symbolic S;
  if (analyze(nV0, nullptr, nE0, from_idx, to_idx, S)) return CGMR_E_INVALID;
  int nV = nV0, next = 0;
  for (int k = 0; k < n_steps; k++) {
    if (nV_step[k] < nV || nE_step[k] < S.nE) return CGMR_E_INVALID;
    Symbolic old = std::move(S);
    if (analyze(nV_step[k], nullptr, nE_step[k], from_idx, to_idx, S, &old)) return CGMR_E_INVALID;
    nV = nV_step[k];
    next += S.extended ? 1 : 0;
  }
  if (n_extended_out) *n_extended_out = next;
  symbolic_info_out(S, nV, out, perm_out);
  return CGMR_OK;
}

int cgmr_marginals(cgmr_ctx* ctx, int nV, const double* poses, const uint8_t* fixed, int nE, const int32_t* ef,
                   const int32_t* et, const double* meas, const double* info, int nK, const int32_t* query,
                   double* cov_out) {
  if (!ctx || !cov_out) return CGMR_E_INVALID;
  return marginal_driver(ctx, 0, nV, poses, fixed, nE, ef, et, meas, info, -1, nK, query, nullptr, nullptr, nullptr, cov_out);
}

int cgmr_covariance_estimate(cgmr_ctx* ctx, int nV, const double* poses, int nE, const int32_t* ef, const int32_t* et,
                             const double* meas, const double* info, int gauge, int nK, const int32_t* query,
                             double* cov_out) {
  if (!ctx || !cov_out) return CGMR_E_INVALID;
  return marginal_driver(ctx, 1, nV, poses, nullptr, nE, ef, et, meas, info, gauge, nK, query, nullptr, nullptr, nullptr, cov_out);
}

int cgmr_condense(cgmr_ctx* ctx, int nV, const double* poses, int nE, const int32_t* ef, const int32_t* et,
                  const double* meas, const double* info, int gauge, int nK, const int32_t* query, int32_t* to_out,
                  double* est_out, double* info_out, double* cov_out) {
  if (!ctx || !to_out || !est_out || !info_out) return CGMR_E_INVALID;
  return marginal_driver(ctx, 2, nV, poses, nullptr, nE, ef, et, meas, info, gauge, nK, query, to_out, est_out, info_out, cov_out);
}

int cgmr_set_symbolic_cache(cgmr_ctx* ctx, int on) {
  if (!ctx) return CGMR_E_INVALID;
  ctx->sym_cache_on = on != 0;
  ctx->sym_valid = false;
  return CGMR_OK;
}

}  // extern "C"
namespace cgmr {
// gn_symbolic.cpp: the helper pool as it runs -- threads an analysis uses (caller included), 1 if the helpers are pinned around
// a last-level cache, the caller's home CPU while it analyses (-1: not pinned), CPUs the process may use, moves of the pool
void host_pool_info(int out[5]);
}
extern "C" {

int cgmr_host_threads_info(int32_t out[5]) {
  if (!out) return CGMR_E_INVALID;
  int v[5];
  host_pool_info(v);
  for (int k = 0; k < 5; k++) out[k] = v[k];
  return CGMR_OK;
}

int cgmr_symbolic_cache_stats(const cgmr_ctx* ctx, int64_t out[2]) {     // the two-value ABI of version 100
  if (!ctx || !out) return CGMR_E_INVALID;
  out[0] = ctx->sym_hits; out[1] = ctx->sym_misses + ctx->sym_extended;
  return CGMR_OK;
}

int cgmr_symbolic_cache_stats3(const cgmr_ctx* ctx, int64_t out[3]) {
  if (!ctx || !out) return CGMR_E_INVALID;
  out[0] = ctx->sym_hits; out[1] = ctx->sym_misses; out[2] = ctx->sym_extended;
  return CGMR_OK;
}

int64_t cgmr_gn_timeouts(const cgmr_ctx* ctx) { return ctx ? ctx->gn_timeouts : -1; }

int cgmr_gn_last_timing(const cgmr_ctx* ctx, double out[5]) {
  if (!ctx || !out) return CGMR_E_INVALID;
  memcpy(out, ctx->timing, sizeof(double) * 5);
  return CGMR_OK;
}

int cgmr_set_profiling(cgmr_ctx* ctx, int on) {
  if (!ctx) return CGMR_E_INVALID;
  ctx->profiling = on != 0;
  memset(ctx->ksec, 0, sizeof ctx->ksec);
  memset(ctx->klaunch, 0, sizeof ctx->klaunch);
  ctx->ev_cls.clear();
  if (on && ctx->ev_pool.empty()) {
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->ev_pool.resize(2 * 2048);              // enough for optimize(~30) on a 21-level tree; further launches go untimed
    for (hipEvent_t& e : ctx->ev_pool)
      if (hipEventCreate(&e) != hipSuccess) { ctx->ev_pool.clear(); return set_err(ctx, CGMR_E_HIP, "hipEventCreate failed"); }
  }
  return CGMR_OK;
}

int cgmr_gn_kernel_times(const cgmr_ctx* ctx, double seconds_out[8], int64_t launches_out[8]) {
  if (!ctx || !seconds_out || !launches_out) return CGMR_E_INVALID;
  memcpy(seconds_out, ctx->ksec, sizeof(double) * 8);
  memcpy(launches_out, ctx->klaunch, sizeof(int64_t) * 8);
  return CGMR_OK;
}

// the same with the classes beyond the first eight: 8 = front_level (k_front_level: a tree level's factorisation and its update
// tiles in one launch); classes 9..11 are reserved (zero)
extern "C" int cgmr_gn_kernel_times_ex(const cgmr_ctx* ctx, double seconds_out[12], int64_t launches_out[12]) {
  if (!ctx || !seconds_out || !launches_out) return CGMR_E_INVALID;
  memcpy(seconds_out, ctx->ksec, sizeof(double) * 12);
  memcpy(launches_out, ctx->klaunch, sizeof(int64_t) * 12);
  return CGMR_OK;
}

}  // extern "C"

// Debugging aid (host only): the front table of a graph's symbolic analysis, 6 ints per front: c0, nc, ns, parent, level, nchild.
extern "C" int cgmr_debug_fronts(int nV, const uint8_t* fixed, int nE, const int32_t* ef, const int32_t* et, int cap, int32_t* out) {
  Symbolic S;
  (void)fixed;
  if (analyze(nV, nullptr, nE, ef, et, S)) return -1;
  int n = (int)S.fronts.size();
  for (int f = 0; f < n && f < cap; f++) {
    const FrontDesc& F = S.fronts[f];
    int32_t* o = out + 6 * f;
    o[0] = F.c0; o[1] = F.nc; o[2] = F.ns; o[3] = F.parent; o[4] = F.level; o[5] = F.nchild;
  }
  return n;
}

// The children's schedule of every front (tests): out[4 f ..] = sched_t, sched_slot, pan_slots, 1 if the front adds into its
// parent's panel (0: root, or child of the top block)
extern "C" int cgmr_debug_schedule(int nV, int nE, const int32_t* ef, const int32_t* et, int cap, int32_t* out) {
  Symbolic S;
  if (analyze(nV, nullptr, nE, ef, et, S)) return -1;
  int n = (int)S.fronts.size();
  for (int f = 0; f < n && f < cap; f++) {
    const FrontDesc& F = S.fronts[f];
    int32_t* o = out + 4 * f;
    o[0] = F.sched_t; o[1] = F.sched_slot; o[2] = F.pan_slots; o[3] = F.ppan_off >= 0 ? 1 : 0;
  }
  return n;
}

// Debugging aid: the (front, chunk) of every work item of the last analysed graph, and each front's parent / level.
extern "C" int cgmr_debug_worklist(const cgmr_ctx* ctx, int32_t* front_out, int32_t* chunk_out, int cap, int32_t* parent_out,
                                   int32_t* level_out, int32_t* ns_out, int fcap) {
  if (!ctx) return -1;
  const Symbolic& S = ctx->sym;
  int n = 0;
  for (int l = 0; l + 1 < (int)S.level_ptr.size(); l++)
    for (int q = S.level_ptr[l]; q < S.level_ptr[l + 1]; q++) {
      int f = S.level_fronts[q];
      int r = 3 * S.fronts[f].ns;
      const int chunk_rows = l < (int)ctx->gn.h_level_chunk.size() ? ctx->gn.h_level_chunk[l] : kChunkRows;
      int nchunk = std::max(1, (r + chunk_rows - 1) / chunk_rows);
      for (int c = 0; c < nchunk; c++) { if (n < cap) { front_out[n] = f; chunk_out[n] = c; } n++; }
    }
  for (int f = 0; f < (int)S.fronts.size() && f < fcap; f++) { parent_out[f] = S.fronts[f].parent; level_out[f] = S.fronts[f].level; ns_out[f] = S.fronts[f].ns; }
  return n;
}

// Host-only test hook: a sequence of complete edge lists analysed one after the other the way a context with the analysis
// cache on does (step k: vertices nV[k], edges e_ptr[k] .. e_ptr[k+1] of ef / et, hub hints h_ptr[k] .. h_ptr[k+1] of hubs).
// out / perm_out / fronts_out (6 ints per front as cgmr_debug_fronts) describe the last analysis; returns its front count.
extern "C" int cgmr_debug_symbolic_steps(int n_steps, const int32_t* nV, const int32_t* e_ptr, const int32_t* ef, const int32_t* et,
                                         const int32_t* h_ptr, const int32_t* hubs, int64_t out[16], int32_t* perm_out,
                                         int32_t* n_extended_out, int fcap, int32_t* fronts_out, int64_t* per_step_out) {
  if (n_steps < 1 || !nV || !e_ptr || !out) return -1;
  Symbolic S;
  std::vector<int32_t> pef, pet;
  int prev_nV = 0, next = 0;
  for (int k = 0; k < n_steps; k++) {
    const int nE = e_ptr[k + 1] - e_ptr[k];
    const int nh = h_ptr ? h_ptr[k + 1] - h_ptr[k] : 0;
    if (analyze_next(S, k > 0, prev_nV, pef, pet, nV[k], nE, ef + e_ptr[k], et + e_ptr[k], nh ? hubs + h_ptr[k] : nullptr, nh)) return -1;
    if (k > 0 && S.extended) next++;
    if (per_step_out) {                                     // per step: levels, extended, ordering us, structure us, factor flops
      int64_t* o = per_step_out + 5 * (size_t)k;
      o[0] = (int64_t)S.level_ptr.size() - 1; o[1] = S.extended ? 1 : 0; o[2] = (int64_t)(1e6 * S.t_order); o[3] = (int64_t)(1e6 * S.t_struct); o[4] = (int64_t)S.flops;
    }
    pef.assign(ef + e_ptr[k], ef + e_ptr[k + 1]);
    pet.assign(et + e_ptr[k], et + e_ptr[k + 1]);
    prev_nV = nV[k];
  }
  if (n_extended_out) *n_extended_out = next;
  symbolic_info_out(S, nV[n_steps - 1], out, perm_out);
  const int n = (int)S.fronts.size();
  for (int f = 0; f < n && f < fcap && fronts_out; f++) {
    const FrontDesc& F = S.fronts[f];
    int32_t* o = fronts_out + 6 * f;
    o[0] = F.c0; o[1] = F.nc; o[2] = F.ns; o[3] = F.parent; o[4] = F.level; o[5] = F.nchild;
  }
  return n;
}

// Tests: the assembly lists (gn_symbolic.h: asm_ptr / asm_src) as the host builds them for an edge list (ctx == nullptr), or
// as they stand on the device for the graph the context analysed last (ef / et ignored).  Returns nf + nb (the number of
// keys; -1: error), the number of list entries in *n_src_out.
extern "C" int cgmr_debug_asm_lists(cgmr_ctx* ctx, int nV, int nE, const int32_t* ef, const int32_t* et, int cap_ptr, int32_t* ptr_out,
                                    int cap_src, int32_t* src_out, int32_t* n_src_out) {
  if (!ptr_out || !src_out || !n_src_out) return -1;
  if (!ctx) {
    Symbolic S;
    if (analyze(nV, nullptr, nE, ef, et, S)) return -1;
    const int nk = S.nf + S.nb;
    if (nk + 1 > cap_ptr || (int)S.asm_src.size() > cap_src) return -1;
    if (S.asm_ptr.empty()) { *n_src_out = 0; return nk; }
    memcpy(ptr_out, S.asm_ptr.data(), 4 * (size_t)(nk + 1));
    memcpy(src_out, S.asm_src.data(), 4 * S.asm_src.size());
    *n_src_out = (int)S.asm_src.size();
    return nk;
  }
  const GnDevice& D = ctx->gn;
  const int nk = D.nf + D.nb;
  if (nk + 1 > cap_ptr || !D.asm_ptr) return -1;
  if (hipSetDevice(ctx->device) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) return -1;
  if (hipMemcpy(ptr_out, D.asm_ptr, 4 * (size_t)(nk + 1), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  const int ns = ptr_out[nk];
  if (ns > cap_src) return -1;
  if (ns > 0 && hipMemcpy(src_out, D.asm_src, 4 * (size_t)ns, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  *n_src_out = ns;
  return nk;
}

// Tests: rel | inv | blk_dst | b_dst (gn_symbolic.h) one behind the other, as the host builds them for an edge list (ctx ==
// nullptr) or as they stand on the device for the graph the context analysed last.  Returns the number of ints, -1: error.
extern "C" int cgmr_debug_maps(cgmr_ctx* ctx, int nV, int nE, const int32_t* ef, const int32_t* et, int cap, int32_t* out) {
  if (!out) return -1;
  if (!ctx) {
    Symbolic S;
    if (analyze(nV, nullptr, nE, ef, et, S)) return -1;
    const size_t n = S.rel.size() + S.inv.size() + S.blk_dst.size() + S.b_dst.size();
    if (n > (size_t)cap) return -1;
    int32_t* o = out;
    for (const std::vector<int32_t>* v : {&S.rel, &S.inv, &S.blk_dst, &S.b_dst}) { memcpy(o, v->data(), 4 * v->size()); o += v->size(); }
    return (int)n;
  }
  const GnDevice& D = ctx->gn;
  const Symbolic& S = ctx->sym;
  const size_t sizes[4] = {(size_t)(S.maps_on_device ? S.n_rel : (int64_t)S.rel.size()), (size_t)(S.maps_on_device ? S.n_inv : (int64_t)S.inv.size()),
                           (size_t)S.nf + S.nb, (size_t)S.nf};
  const int32_t* srcs[4] = {D.rel, D.inv, D.blk_dst, D.b_dst};
  if (sizes[0] + sizes[1] + sizes[2] + sizes[3] > (size_t)cap) return -1;
  if (hipSetDevice(ctx->device) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) return -1;
  int32_t* o = out;
  for (int k = 0; k < 4; k++) {
    if (sizes[k] && hipMemcpy(o, srcs[k], 4 * sizes[k], hipMemcpyDeviceToHost) != hipSuccess) return -1;
    o += sizes[k];
  }
  return (int)(o - out);
}
