// Native cross-clip tracker: the HOST half of the reference's OverTracker
// (mdqe/tracking/OverTracker.py:10-242) as a C++ object behind the C ABI, plus the entry points that
// drive the device half (tracker.hip) for one clip or a whole run of clips without returning to Python.
//
// State split: the device holds a running per-(instance, frame) SUM of mask logits and a
// count; everything that is O(instances x 256) lives here: embeddings, class probabilities, presence
// flags, untracked counters, the bi-softmax similarity (:228-242), the rectangular assignment
// (scipy.optimize.linear_sum_assignment, :159 -- the reference's third-party dependency, scipy==1.8.1 in
// requirements.txt:3; its published algorithm, Crouse's shortest augmenting path for rectangular costs,
// is restated in lsap_solve below and pinned against scipy itself in tests/test_tracker_native_cpu.py)
// and the ID bookkeeping (:115-193).
#include "common.h"

#include <algorithm>
#include <cmath>
#include <time.h>
#include <cstdlib>
#include <limits>
#include <set>
#include <vector>

// ------------------------------------------------------------------------------------------------
// scipy.optimize.linear_sum_assignment (rectangular_lsap): same traversal order and tie-breaking
// ------------------------------------------------------------------------------------------------
namespace {

long lsap_augment(long nc, const std::vector<double>& cost, std::vector<double>& u, std::vector<double>& v,
                  std::vector<long>& path, std::vector<long>& row4col, std::vector<double>& spc, long i,
                  std::vector<char>& SR, std::vector<char>& SC, std::vector<long>& remaining, double* p_min) {
  double min_val = 0;
  long num_remaining = nc;
  for (long it = 0; it < nc; ++it) remaining[it] = nc - it - 1;     // filled in reverse, as scipy does
  std::fill(SR.begin(), SR.end(), 0);
  std::fill(SC.begin(), SC.end(), 0);
  std::fill(spc.begin(), spc.end(), std::numeric_limits<double>::infinity());
  long sink = -1;
  while (sink == -1) {
    long index = -1;
    double lowest = std::numeric_limits<double>::infinity();
    SR[i] = 1;
    for (long it = 0; it < num_remaining; ++it) {
      const long j = remaining[it];
      const double r = min_val + cost[i * nc + j] - u[i] - v[j];
      if (r < spc[j]) { path[j] = i; spc[j] = r; }
      // several columns at the minimum: prefer one that is a new sink
      if (spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1)) { lowest = spc[j]; index = it; }
    }
    min_val = lowest;
    if (min_val == std::numeric_limits<double>::infinity()) return -1;
    const long j = remaining[index];
    if (row4col[j] == -1) sink = j; else i = row4col[j];
    SC[j] = 1;
    remaining[index] = remaining[--num_remaining];
  }
  *p_min = min_val;
  return sink;
}

// cost [nr, nc] row-major.  Fills rows/cols (min(nr, nc) pairs, ascending rows).  Returns 0, or -1 if infeasible.
int lsap_solve(long nr, long nc, const double* cost_in, bool maximize, std::vector<long>& rows, std::vector<long>& cols) {
  rows.clear(); cols.clear();
  if (nr == 0 || nc == 0) return 0;
  const bool transpose = nc < nr;
  std::vector<double> cost((size_t)nr * nc);
  if (transpose) {
    for (long i = 0; i < nr; ++i) for (long j = 0; j < nc; ++j) cost[j * nr + i] = cost_in[i * nc + j];
    std::swap(nr, nc);
  } else {
    std::copy(cost_in, cost_in + (size_t)nr * nc, cost.begin());
  }
  if (maximize) for (auto& c : cost) c = -c;
  for (auto c : cost) if (c != c || c == -std::numeric_limits<double>::infinity()) return -1;
  std::vector<double> u(nr, 0), v(nc, 0), spc(nc);
  std::vector<long> path(nc, -1), col4row(nr, -1), row4col(nc, -1), remaining(nc);
  std::vector<char> SR(nr), SC(nc);
  for (long cur = 0; cur < nr; ++cur) {
    double min_val;
    const long sink = lsap_augment(nc, cost, u, v, path, row4col, spc, cur, SR, SC, remaining, &min_val);
    if (sink < 0) return -1;
    u[cur] += min_val;
    for (long i = 0; i < nr; ++i) if (SR[i] && i != cur) u[i] += min_val - spc[col4row[i]];
    for (long j = 0; j < nc; ++j) if (SC[j]) v[j] -= min_val - spc[j];
    long j = sink;
    while (true) {
      const long i = path[j];
      row4col[j] = i;
      std::swap(col4row[i], j);
      if (i == cur) break;
    }
  }
  if (transpose) {                     // pairs (col4row[i], i) ordered by the original row = col4row[i]
    std::vector<long> order(nr);
    for (long i = 0; i < nr; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](long a, long b) { return col4row[a] < col4row[b]; });
    for (long i : order) { rows.push_back(col4row[i]); cols.push_back(i); }
  } else {
    for (long i = 0; i < nr; ++i) { rows.push_back(i); cols.push_back(col4row[i]); }
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// bi-softmax similarity, OverTracker.py:228-242.  saved [ns, E] (row list), inp [ni, E] -> out [ns, ni]
// ------------------------------------------------------------------------------------------------
void ctt_similarity(const std::vector<const float*>& saved, const float* inp, long ni, long E, std::vector<float>& out) {
  const long ns = (long)saved.size();
  out.assign((size_t)ns * ni, 0.f);
  if (ns == 0 || ni == 0) return;
  std::vector<float> f((size_t)ns * ni);
  for (long s = 0; s < ns; ++s)
    for (long j = 0; j < ni; ++j) {
      double acc = 0;
      const float* a = saved[s];
      const float* b = inp + j * E;
      for (long e = 0; e < E; ++e) acc += (double)a[e] * (double)b[e];
      f[s * ni + j] = (float)acc;
    }
  std::vector<float> d2t((size_t)ns * ni), t2d((size_t)ns * ni);
  for (long j = 0; j < ni; ++j) {                                   // softmax over the saved axis
    float m = f[j];
    for (long s = 1; s < ns; ++s) m = std::max(m, f[s * ni + j]);
    double sum = 0;
    for (long s = 0; s < ns; ++s) sum += std::exp((double)(f[s * ni + j] - m));
    for (long s = 0; s < ns; ++s) d2t[s * ni + j] = (float)(std::exp((double)(f[s * ni + j] - m)) / sum);
  }
  for (long s = 0; s < ns; ++s) {                                   // softmax over the input axis
    float m = f[s * ni];
    for (long j = 1; j < ni; ++j) m = std::max(m, f[s * ni + j]);
    double sum = 0;
    for (long j = 0; j < ni; ++j) sum += std::exp((double)(f[s * ni + j] - m));
    for (long j = 0; j < ni; ++j) t2d[s * ni + j] = (float)(std::exp((double)(f[s * ni + j] - m)) / sum);
  }
  const float Ws = ns > 1 ? 1.f : 0.f, Wi = ni > 1 ? 1.f : 0.f;
  if (ns == 1 && ni == 1) { out[0] = 0.5f * (d2t[0] + t2d[0]); return; }
  const float den = std::max(Ws + Wi, 1.f);
  for (size_t k = 0; k < out.size(); ++k) out[k] = (Ws * d2t[k] + Wi * t2d[k]) / den;
}

struct Tracker {
  int max_inst, T, win, stride, K, E;
  float thr;
  int num_inst = 0, mem_len, num_clips, start_frame = 0, num_clip = 0, n_long, n_short;
  std::set<int> saved_idx;
  std::vector<float> cnt_h, cls, embeds, w_mem, untracked, embed_mem;
  std::vector<char> clip_valid;
  // scratch of the last decide()
  std::vector<int> r_idx, c_idx;

  Tracker(int max_inst_, int T_, int win_, int stride_, int K_, int E_, float thr_)
      : max_inst(max_inst_), T(T_), win(win_), stride(stride_), K(K_), E(E_), thr(thr_) {
    mem_len = win + T;
    num_clips = win / stride + 2;
    n_long = 15 / stride;
    n_short = std::max(T, 5) / stride;
    cnt_h.assign((size_t)max_inst * mem_len, 0.f);
    clip_valid.assign((size_t)num_clips * max_inst, 0);
    cls.assign((size_t)num_clips * max_inst * K, 0.f);
    embeds.assign((size_t)num_clips * max_inst * E, 0.f);
    w_mem.resize(std::max(std::max(n_long, T), 3));
    for (int i = 0; i < (int)w_mem.size(); ++i) w_mem[i] = (float)std::exp((double)i * 0.25);     // exp(0.25 i), correctly rounded
    untracked.assign(max_inst, 0.f);
    embed_mem.assign((size_t)max_inst * E, 0.f);
  }

  // ---- the counts' fast path (round 5, tracker.hip trk_siou_host_kernel): device accumulator + ticket (kept zero between launches) and
  // host-coherent pinned memory for the finished counts + a sequence flag, owned by the object and created on the first update that
  // needs counts (mdqe_tracker_create makes no HIP call: the CPU tests drive the host half alone).  MDQE_TRK_FAST=0 in the
  // environment (or mdqe_debug_trk_fast(0)) keeps the four-step form (memset, kernel, copy, synchronize) -- same counts, same decisions.
  float* acc_dev = nullptr;
  unsigned* ticket_dev = nullptr;
  unsigned long long* words_pin = nullptr;   // (launch sequence number << 32 | count bits), written by the kernel
  std::vector<float> counts_fast;            // the unpacked counts of the last launch
  long acc_cap = 0;
  unsigned seq = 0;

  int ensure_fast(long need, hipStream_t st) {
    if (need <= acc_cap) return MDQE_OK;
    if (acc_dev != nullptr && hipStreamSynchronize(st) != hipSuccess) return MDQE_ELAUNCH;     // (a kernel may still use the old buffers)
    release_fast();
    long cap = 4096;
    while (cap < need) cap *= 2;
    if (hipMalloc((void**)&acc_dev, (size_t)cap * sizeof(float)) != hipSuccess) { acc_dev = nullptr; return MDQE_ELAUNCH; }
    if (hipMalloc((void**)&ticket_dev, 64) != hipSuccess) { release_fast(); return MDQE_ELAUNCH; }
    if (hipHostMalloc((void**)&words_pin, (size_t)cap * sizeof(unsigned long long), hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) { words_pin = nullptr; release_fast(); return MDQE_ELAUNCH; }
    if (hipMemsetAsync(acc_dev, 0, (size_t)cap * sizeof(float), st) != hipSuccess || hipMemsetAsync(ticket_dev, 0, 64, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) { release_fast(); return MDQE_ELAUNCH; }
    for (long k = 0; k < cap; ++k) words_pin[k] = 0ull;
    counts_fast.assign((size_t)cap, 0.f);
    acc_cap = cap;
    return MDQE_OK;
  }

  void release_fast() {
    if (acc_dev) (void)hipFree(acc_dev);
    if (ticket_dev) (void)hipFree(ticket_dev);
    if (words_pin) (void)hipHostFree(words_pin);
    acc_dev = nullptr; ticket_dev = nullptr; words_pin = nullptr; acc_cap = 0;
  }

  ~Tracker() { release_fast(); }

  // frames of the clip [f0, f0+nf_clip) already saved in this window: bank frame s0, clip frame a, count
  // set when a call failed AFTER part of its work reached the device bank (a kernel / copy error in the middle of an update or
  // of a run of clips): host and device halves may disagree from then on, so every later call is refused (MDQE_ESTATE).
  bool poisoned = false;

  int overlap(int f0, int nf_clip, int* s0, int* a, int* nf) const {
    if (poisoned) return MDQE_ESTATE;
    int first = -1, lastf = -1, count = 0;
    for (int o = 0; o < nf_clip; ++o) {
      const int f = f0 + o;
      if (f >= start_frame && saved_idx.count(f)) {
        if (f - start_frame >= mem_len) return MDQE_EINVAL;
        if (first < 0) first = o;
        lastf = o; ++count;
      }
    }
    if (count > 0 && lastf - first + 1 != count) return MDQE_EINVAL;   // overlapping frames must be contiguous (stride <= T)
    *nf = count; *a = count ? first : 0; *s0 = count ? f0 + first - start_frame : 0;
    return MDQE_OK;
  }

  // OverTracker.update (:115-193) given the (inter, |saved|, |input|) counts of the overlapping frames.
  // Fills r_idx / c_idx (bank row <- clip instance) and the frame range of the memory write.
  int decide(int f0, int nf_clip, int n_in, const float* scores, const float* cls_probs, const float* emb,
             const float* counts3, int have_counts, int* s0_out, int* a_out, int* nf_out) {
    r_idx.clear(); c_idx.clear();
    // everything that can refuse the clip is checked BEFORE any state changes (the object is reused after an error)
    if (poisoned) return MDQE_ESTATE;
    if (num_clip >= num_clips) return MDQE_EINVAL;
    const int s0 = std::max(f0 - start_frame, 0);
    const int s1 = f0 + nf_clip - 1 - start_frame;
    if (s1 >= mem_len || s1 < s0) return MDQE_EINVAL;
    std::vector<float> siou, sm;
    int ni = 0;
    int n_first = 0;                                  // rows created by the very first clip (all of its instances)
    std::vector<char> matched(n_in, 0);
    if (num_inst == 0) {
      if (n_in > max_inst) return MDQE_EINVAL;
      for (int i = 0; i < n_in; ++i) { r_idx.push_back(i); c_idx.push_back(i); matched[i] = 1; }
      n_first = n_in;
    } else {
      ni = num_inst;
      std::vector<int> lo, sh;
      for (int i = 0; i < ni; ++i) {
        if (untracked[i] < (float)n_long) lo.push_back(i);
        if (untracked[i] < (float)n_short) sh.push_back(i);
      }
      sm.assign((size_t)ni * n_in, 0.f);
      std::vector<const float*> rows;
      std::vector<float> tmp;
      for (int i : lo) rows.push_back(&embed_mem[(size_t)i * E]);
      ctt_similarity(rows, emb, n_in, E, tmp);
      for (size_t k = 0; k < lo.size(); ++k) std::copy(tmp.begin() + k * n_in, tmp.begin() + (k + 1) * n_in, sm.begin() + (size_t)lo[k] * n_in);
      if (!(sh.size() == lo.size() && n_short <= n_long)) {      // same index set: 0.5 * (a + a) == a bit for bit
        rows.clear();
        for (int i : sh) rows.push_back(&embed_mem[(size_t)i * E]);
        ctt_similarity(rows, emb, n_in, E, tmp);
        for (size_t k = 0; k < sh.size(); ++k)
          for (int j = 0; j < n_in; ++j) {
            float& d = sm[(size_t)sh[k] * n_in + j];
            d = 0.5f * (d + tmp[k * n_in + j]);
          }
      }
      siou.assign((size_t)ni * n_in, 0.f);
      if (have_counts && n_in > 0) {
        for (size_t k = 0; k < siou.size(); ++k) {
          const float inter = counts3[3 * k], sa = counts3[3 * k + 1], ia = counts3[3 * k + 2];
          siou[k] = (sa > 0.f && ia > 0.f) ? inter / (((sa + ia) - inter) + 1e-6f) : 0.f;
        }
      }
      std::vector<double> sc((size_t)ni * n_in);
      std::vector<char> above((size_t)ni * n_in);
      for (size_t k = 0; k < sc.size(); ++k) {
        const float s = siou[k] + sm[k];
        above[k] = s > 0.6f;
        sc[k] = above[k] ? (double)s : 0.0;
      }
      std::vector<long> rr, cc;
      if (lsap_solve(ni, n_in, sc.data(), true, rr, cc) != 0) return MDQE_EINVAL;
      for (size_t k = 0; k < rr.size(); ++k) {
        const size_t at = (size_t)rr[k] * n_in + cc[k];
        if (!above[at]) continue;
        r_idx.push_back((int)rr[k]); c_idx.push_back((int)cc[k]); matched[cc[k]] = 1;
        siou[at] = -1.f; sm[at] = 0.f;
      }
    }
    // unmatched detections: duplicates of a saved instance are dropped, confident ones become new IDs (:171-185)
    std::vector<int> fresh;
    for (int i = 0; i < n_in; ++i) {
      if (matched[i]) continue;
      bool rep = false;
      if (ni > 0) {
        float ms = -std::numeric_limits<float>::infinity(), mc = ms;
        for (int s = 0; s < ni; ++s) { ms = std::max(ms, siou[(size_t)s * n_in + i]); mc = std::max(mc, sm[(size_t)s * n_in + i]); }
        rep = ms > 0.4f || mc > 0.6f;
      }
      if (!rep && scores[i] > (float)(2.0 * (double)thr)) fresh.push_back(i);
    }
    const int n_new = (int)fresh.size();
    if (num_inst + n_first + n_new > max_inst) { r_idx.clear(); c_idx.clear(); return MDQE_EINVAL; }   // (nothing committed yet)
    for (int k = 0; k < n_new; ++k) { r_idx.push_back(num_inst + k); c_idx.push_back(fresh[k]); }
    // _update_memory (:65-90) -- from here on the clip is committed
    const int a = start_frame + s0 - f0;
    *s0_out = s0; *a_out = a; *nf_out = s1 - s0 + 1;
    for (auto& x : untracked) x += 1.f;
    const int nc = num_clip;
    for (size_t k = 0; k < r_idx.size(); ++k) {
      const int r = r_idx[k], c = c_idx[k];
      for (int f = s0; f <= s1; ++f) cnt_h[(size_t)r * mem_len + f] += 1.f;
      clip_valid[(size_t)nc * max_inst + r] = 1;
      std::copy(cls_probs + (size_t)c * K, cls_probs + (size_t)(c + 1) * K, &cls[((size_t)nc * max_inst + r) * K]);
      std::copy(emb + (size_t)c * E, emb + (size_t)(c + 1) * E, &embeds[((size_t)nc * max_inst + r) * E]);
      untracked[r] = 0.f;
      float* em = &embed_mem[(size_t)r * E];
      if (nc > 0) {                                    // exp-weighted mean of the last <= 3 clip embeddings (:81-88)
        const int st = std::max(nc - 2, 0);
        float wsum = 0.f;
        std::vector<float> acc(E, 0.f);
        for (int q = st; q <= nc; ++q) {
          const float* e = &embeds[((size_t)q * max_inst + r) * E];
          const float w = w_mem[q - st];
          bool any = false;
          for (int d = 0; d < E; ++d) { acc[d] += e[d] * w; any = any || e[d] != 0.f; }
          wsum += any ? w : 0.f;
        }
        const float den = std::max(wsum, 1.f);
        for (int d = 0; d < E; ++d) em[d] = acc[d] / den;
      } else {
        std::copy(emb + (size_t)c * E, emb + (size_t)(c + 1) * E, em);
      }
    }
    for (int o = 0; o < nf_clip; ++o) saved_idx.insert(f0 + o);
    num_clip += 1;
    num_inst += n_first + n_new;
    return MDQE_OK;
  }

  // OverTracker.get_result (:195-225), host part.  out_cls [n, K]; carry_valid [n, mem_len - win] (only when !is_last).
  int result(int is_last, float* out_cls, int* n_out, int* ln_out, unsigned char* carry_valid) {
    if (poisoned) return MDQE_ESTATE;
    const int n = num_inst;
    if (saved_idx.empty()) return MDQE_EINVAL;
    const int nv = *saved_idx.rbegin() - start_frame + 1;
    *n_out = n;
    *ln_out = is_last ? nv : win;
    std::vector<float> oc((size_t)n * K, 0.f), oq((size_t)n * E, 0.f);
    const int nc = std::min(std::max(3, (T - 1) / stride), num_clip);
    for (int i = 0; i < n; ++i) {
      float cnt = 0.f;
      for (int q = 0; q < num_clip; ++q) {
        const float v = clip_valid[(size_t)q * max_inst + i] ? 1.f : 0.f;
        cnt += v;
        const float* c = &cls[((size_t)q * max_inst + i) * K];
        for (int k = 0; k < K; ++k) oc[(size_t)i * K + k] += c[k] * v;
      }
      const float den = std::max(cnt, 1.f);
      for (int k = 0; k < K; ++k) oc[(size_t)i * K + k] /= den;
      float wsum = 0.f;
      for (int q = num_clip - nc; q < num_clip; ++q) {
        const float w = (clip_valid[(size_t)q * max_inst + i] ? 1.f : 0.f) * w_mem[q - (num_clip - nc)];
        wsum += w;
        const float* e = &embeds[((size_t)q * max_inst + i) * E];
        for (int d = 0; d < E; ++d) oq[(size_t)i * E + d] += e[d] * w;
      }
      const float dq = std::max(wsum, 1.f);
      for (int d = 0; d < E; ++d) oq[(size_t)i * E + d] /= dq;
    }
    std::copy(oc.begin(), oc.end(), out_cls);
    if (!is_last) {
      const int k = mem_len - win;
      std::vector<char> cv((size_t)n * k);
      for (int i = 0; i < n; ++i)
        for (int f = 0; f < k; ++f) cv[(size_t)i * k + f] = cnt_h[(size_t)i * mem_len + win + f] > 0.f;
      // _init_memory(False)
      num_clip = 1;
      start_frame += win;
      saved_idx.erase(saved_idx.begin(), saved_idx.lower_bound(start_frame));
      std::fill(cnt_h.begin(), cnt_h.end(), 0.f);
      std::fill(clip_valid.begin(), clip_valid.end(), 0);
      std::fill(cls.begin(), cls.end(), 0.f);
      std::fill(embeds.begin(), embeds.end(), 0.f);
      for (int i = 0; i < n; ++i) {
        bool any = false;
        for (int f = 0; f < k; ++f) {
          const char v = cv[(size_t)i * k + f];
          cnt_h[(size_t)i * mem_len + f] = v ? 1.f : 0.f;
          if (carry_valid) carry_valid[(size_t)i * k + f] = (unsigned char)v;
          any = any || v;
        }
        clip_valid[i] = any;
        std::copy(oc.begin() + (size_t)i * K, oc.begin() + (size_t)(i + 1) * K, &cls[(size_t)i * K]);
        std::copy(oq.begin() + (size_t)i * E, oq.begin() + (size_t)(i + 1) * E, &embeds[(size_t)i * E]);
      }
    }
    return MDQE_OK;
  }
};

}  // namespace

// device half (tracker.hip)
extern "C" int mdqe_trk_siou_f32(const float*, long, int, const float*, long, int, long, float*, void*);
extern "C" int mdqe_trk_siou_host_f32(const float*, long, int, const float*, long, int, long, float*, unsigned*, unsigned long long*, unsigned, void*);
extern "C" int mdqe_trk_wait_counts(const unsigned long long*, int, unsigned, int, float*, void*);
extern "C" int mdqe_trk_accumulate_f32(float*, long, float*, long, const float*, long, long, int, const int*, const int*, int, void*);
extern "C" int mdqe_trk_window_mean_f32(const float*, const float*, long, int, int, int, long, float*, void*);
extern "C" int mdqe_trk_carry_f32(float*, float*, long, int, int, int, long, float*, void*);

extern "C" int mdqe_lsap_f64(const double* cost, int nr, int nc, int maximize, int* rows_out, int* cols_out, int* n_out) {
  MDQE_REQUIRE(nr >= 0 && nc >= 0);
  MDQE_CHECK_PTR(n_out);
  if (nr > 0 && nc > 0) { MDQE_CHECK_PTR(cost); MDQE_CHECK_PTR(rows_out); MDQE_CHECK_PTR(cols_out); }
  std::vector<long> r, c;
  if (lsap_solve(nr, nc, cost, maximize != 0, r, c) != 0) return MDQE_EINVAL;
  *n_out = (int)r.size();
  for (size_t k = 0; k < r.size(); ++k) { rows_out[k] = (int)r[k]; cols_out[k] = (int)c[k]; }
  return MDQE_OK;
}

extern "C" int mdqe_tracker_create(int max_inst, int T, int win, int stride, int K, int E, float thr, void** handle) {
  MDQE_REQUIRE(max_inst > 0 && T > 0 && win > 0 && stride > 0 && K > 0 && E > 0);
  MDQE_CHECK_PTR(handle);
  *handle = new Tracker(max_inst, T, win, stride, K, E, thr);
  return MDQE_OK;
}

extern "C" int mdqe_tracker_destroy(void* handle) {
  delete static_cast<Tracker*>(handle);
  return MDQE_OK;
}

extern "C" int mdqe_tracker_state(void* handle, int* num_inst, int* num_clip, int* start_frame) {
  MDQE_CHECK_PTR(handle);
  Tracker* t = static_cast<Tracker*>(handle);
  if (num_inst) *num_inst = t->num_inst;
  if (num_clip) *num_clip = t->num_clip;
  if (start_frame) *start_frame = t->start_frame;
  return MDQE_OK;
}

extern "C" int mdqe_tracker_overlap(void* handle, int f0, int n_frames, int* ni, int* s0, int* a, int* nf) {
  MDQE_CHECK_PTR(handle); MDQE_CHECK_PTR(ni); MDQE_CHECK_PTR(s0); MDQE_CHECK_PTR(a); MDQE_CHECK_PTR(nf);
  Tracker* t = static_cast<Tracker*>(handle);
  *ni = t->num_inst;
  if (t->num_inst == 0) { *s0 = *a = *nf = 0; return MDQE_OK; }
  return t->overlap(f0, n_frames, s0, a, nf);
}

extern "C" int mdqe_tracker_decide(void* handle, int f0, int n_frames, int n_in, const float* scores, const float* cls_probs,
                                   const float* embeds, const float* counts3, int* r_idx, int* c_idx, int* n_pairs,
                                   int* s0, int* a, int* nf) {
  MDQE_CHECK_PTR(handle); MDQE_CHECK_PTR(n_pairs); MDQE_CHECK_PTR(s0); MDQE_CHECK_PTR(a); MDQE_CHECK_PTR(nf);
  MDQE_REQUIRE(n_frames > 0 && n_in >= 0);
  if (n_in > 0) { MDQE_CHECK_PTR(scores); MDQE_CHECK_PTR(cls_probs); MDQE_CHECK_PTR(embeds); MDQE_CHECK_PTR(r_idx); MDQE_CHECK_PTR(c_idx); }
  Tracker* t = static_cast<Tracker*>(handle);
  const int rc = t->decide(f0, n_frames, n_in, scores, cls_probs, embeds, counts3, counts3 != nullptr, s0, a, nf);
  if (rc != MDQE_OK) return rc;
  *n_pairs = (int)t->r_idx.size();
  for (size_t k = 0; k < t->r_idx.size(); ++k) { r_idx[k] = t->r_idx[k]; c_idx[k] = t->c_idx[k]; }
  return MDQE_OK;
}

extern "C" int mdqe_tracker_result(void* handle, int is_last, float* out_cls, int* n, int* ln, unsigned char* carry_valid) {
  MDQE_CHECK_PTR(handle); MDQE_CHECK_PTR(out_cls); MDQE_CHECK_PTR(n); MDQE_CHECK_PTR(ln);
  return static_cast<Tracker*>(handle)->result(is_last, out_cls, n, ln, carry_valid);
}

// One tracker update with the device half included: sign-intersection counts of the overlapping frames (kernel +
// one small D2H + a stream sync: the assignment needs them), the decisions, the indexed accumulate.
// bank_sum [max_inst, mem_len, hw], bank_cnt [max_inst, mem_len] (device); masks [n_in, mask_frames, hw] device fp32, rows
// inst_stride floats apart; counts_dev / counts_host: scratch of >= max_inst*n_in*3 floats (device / pinned host).
static int env_int(const char* name, int dflt) { const char* v = getenv(name); return (v && *v) ? atoi(v) : dflt; }
static int g_trk_fast = env_int("MDQE_TRK_FAST", 1);          // 0: the counts by memset + kernel + copy + synchronize (rounds 1-4)
static int g_trk_spin_us = env_int("MDQE_TRK_SPIN_US", 2000);  // how long the host polls the flag before it falls back to a stream synchronize
extern "C" int mdqe_debug_trk_fast(int v) { g_trk_fast = v; return MDQE_OK; }
extern "C" int mdqe_debug_trk_spin_us(int v) { g_trk_spin_us = v; return MDQE_OK; }

// tools/replay_profile.py: where the host time of an update goes -- [0] launching the counts kernel, [1] waiting for the counts, [2] the host
// decision, [3] launching the accumulate kernel, [4] updates; seconds, accumulated while mdqe_debug_trk_times(..., reset) has switched it on
static int g_trk_timing = 0;
static double g_trk_t[5] = {0, 0, 0, 0, 0};
static inline double trk_now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
#define TRK_T0() double trk_t_ = g_trk_timing ? trk_now() : 0.0
#define TRK_T(k) do { if (g_trk_timing) { const double n_ = trk_now(); g_trk_t[k] += n_ - trk_t_; trk_t_ = n_; } } while (0)
extern "C" int mdqe_debug_trk_times(double* out5, int on) {
  if (out5) for (int k = 0; k < 5; ++k) out5[k] = g_trk_t[k];
  for (int k = 0; k < 5; ++k) g_trk_t[k] = 0;
  g_trk_timing = on;
  return MDQE_OK;
}

static int tracker_update_one(Tracker* t, float* bank_sum, float* bank_cnt, long hw, int f0, int n_frames, int n_in,
                              const float* scores, const float* cls_probs, const float* embeds, const float* masks,
                              long inst_stride, float* counts_dev, float* counts_host, hipStream_t st) {
  int s0 = 0, a = 0, nf = 0;
  const int ni = t->num_inst;
  const float* c3 = nullptr;
  if (ni > 0) {
    const int rc = t->overlap(f0, n_frames, &s0, &a, &nf);
    if (rc != MDQE_OK) return rc;
    if (nf > 0 && n_in > 0) {
      const long bank_stride = (long)t->mem_len * hw;
      if (g_trk_fast) {
        // one kernel: counts -> host-coherent memory + a sequence flag; the host polls the flag (tracker.hip)
        int rc2 = t->ensure_fast((long)ni * n_in * 3, st);
        if (rc2 != MDQE_OK) return rc2;
        if (++t->seq == 0) t->seq = 1;                                   // (0 marks a word that was never written)
        const unsigned seq = t->seq;
        TRK_T0();
        rc2 = mdqe_trk_siou_host_f32(bank_sum + (long)s0 * hw, bank_stride, ni, masks + (long)a * hw, inst_stride, n_in, (long)nf * hw,
                                     t->acc_dev, t->ticket_dev, t->words_pin, seq, st);
        if (rc2 != MDQE_OK) return rc2;                                  // (read-only so far: the tracker is still consistent)
        TRK_T(0);
        rc2 = mdqe_trk_wait_counts(t->words_pin, ni * n_in * 3, seq, g_trk_spin_us, t->counts_fast.data(), st);    // the one host wait of an update
        if (rc2 != MDQE_OK) { t->poisoned = true; return rc2; }          // (the accumulator / ticket may be left dirty)
        TRK_T(1);
        c3 = t->counts_fast.data();
      } else {
        TRK_T0();
        int rc2 = mdqe_trk_siou_f32(bank_sum + (long)s0 * hw, bank_stride, ni, masks + (long)a * hw, inst_stride, n_in, (long)nf * hw,
                                    counts_dev, st);
        if (rc2 != MDQE_OK) return rc2;                                  // (read-only so far: the tracker is still consistent)
        if (hipMemcpyAsync(counts_host, counts_dev, (size_t)ni * n_in * 3 * sizeof(float), hipMemcpyDeviceToHost, st) != hipSuccess)
          return MDQE_ELAUNCH;
        TRK_T(0);
        if (hipStreamSynchronize(st) != hipSuccess) return MDQE_ELAUNCH;   // the one host sync of an update
        TRK_T(1);
        c3 = counts_host;
      }
    }
  }
  double trk_t_ = g_trk_timing ? trk_now() : 0.0;
  const int rc = t->decide(f0, n_frames, n_in, scores, cls_probs, embeds, c3, c3 != nullptr, &s0, &a, &nf);
  if (rc != MDQE_OK) return rc;                                          // (decide validates before it commits)
  TRK_T(2);
  if (g_trk_timing) g_trk_t[4] += 1;
  if (!t->r_idx.empty()) {
    const int rc3 = mdqe_trk_accumulate_f32(bank_sum + (long)s0 * hw, (long)t->mem_len * hw, bank_cnt + s0, t->mem_len, masks + (long)a * hw,
                                            inst_stride, (long)nf * hw, nf, t->r_idx.data(), t->c_idx.data(), (int)t->r_idx.size(), st);
    if (rc3 != MDQE_OK) t->poisoned = true;                              // host half committed, device half not
    TRK_T(3);
    return rc3;
  }
  return MDQE_OK;
}

extern "C" int mdqe_tracker_update(void* handle, float* bank_sum, float* bank_cnt, long hw, int f0, int n_frames, int n_in,
                                   const float* scores, const float* cls_probs, const float* embeds, const float* masks,
                                   long inst_stride, float* counts_dev, float* counts_host, void* stream) {
  MDQE_CHECK_PTR(handle); MDQE_CHECK_PTR(bank_sum); MDQE_CHECK_PTR(bank_cnt); MDQE_CHECK_PTR(counts_dev); MDQE_CHECK_PTR(counts_host);
  MDQE_REQUIRE(hw > 0 && hw % 4 == 0 && n_frames > 0 && n_in >= 0 && inst_stride % 4 == 0);
  if (n_in > 0) { MDQE_CHECK_PTR(scores); MDQE_CHECK_PTR(cls_probs); MDQE_CHECK_PTR(embeds); MDQE_CHECK_PTR(masks); }
  return tracker_update_one(static_cast<Tracker*>(handle), bank_sum, bank_cnt, hw, f0, n_frames, n_in, scores, cls_probs, embeds,
                            masks, inst_stride, counts_dev, counts_host, (hipStream_t)stream);
}

// A run of clips in one call (rank 0's replay of a gathered round: no Python between clips).  Clip i: frames
// [f0[i], f0[i]+n_frames[i]), n_in[i] instances whose host vectors start at row row0[i] of scores / cls_probs / embeds
// and whose device masks start at masks[i] (HOST array of device pointers).
extern "C" int mdqe_tracker_update_many(void* handle, float* bank_sum, float* bank_cnt, long hw, int n_clips, const int* f0,
                                        const int* n_frames, const int* n_in, const int* row0, const float* scores,
                                        const float* cls_probs, const float* embeds, const float* const* masks,
                                        const long* inst_stride, float* counts_dev, float* counts_host, void* stream) {
  MDQE_CHECK_PTR(handle); MDQE_CHECK_PTR(bank_sum); MDQE_CHECK_PTR(bank_cnt); MDQE_CHECK_PTR(counts_dev); MDQE_CHECK_PTR(counts_host);
  MDQE_REQUIRE(hw > 0 && hw % 4 == 0 && n_clips >= 0);
  if (n_clips == 0) return MDQE_OK;
  MDQE_CHECK_PTR(f0); MDQE_CHECK_PTR(n_frames); MDQE_CHECK_PTR(n_in); MDQE_CHECK_PTR(row0); MDQE_CHECK_PTR(masks); MDQE_CHECK_PTR(inst_stride);
  Tracker* t = static_cast<Tracker*>(handle);
  if (t->poisoned) return MDQE_ESTATE;
  for (int i = 0; i < n_clips; ++i) {                                   // argument checks for the WHOLE run before its first clip is committed
    MDQE_REQUIRE(n_frames[i] > 0 && n_in[i] >= 0 && inst_stride[i] % 4 == 0 && row0[i] >= 0);
    if (n_in[i] > 0) { MDQE_CHECK_PTR(scores); MDQE_CHECK_PTR(cls_probs); MDQE_CHECK_PTR(embeds); MDQE_CHECK_PTR(masks[i]); }
  }
  for (int i = 0; i < n_clips; ++i) {
    const long o = row0[i];
    const int rc = tracker_update_one(t, bank_sum, bank_cnt, hw, f0[i], n_frames[i], n_in[i], scores ? scores + o : nullptr,
                                      cls_probs ? cls_probs + o * t->K : nullptr, embeds ? embeds + o * t->E : nullptr, masks[i],
                                      inst_stride[i], counts_dev, counts_host, (hipStream_t)stream);
    if (rc != MDQE_OK) {
      if (i > 0) t->poisoned = true;                                    // earlier clips of the run are committed: the run is not atomic, so
      return rc;                                                        // the object refuses further use instead of going on half-updated
    }
  }
  return MDQE_OK;
}

// get_result with the device half: out_masks [n, ln, hw] = per-frame mean logits of the window; unless is_last the
// bank is re-based (the last mem_len - win frames become its first ones, :209-225).  carry: device scratch of
// >= max_inst*(mem_len-win)*hw floats.  out_cls [n, K] host.  *n / *ln tell the caller how much of out_masks was written
// (its capacity must be max_inst*max(win, mem_len)*hw floats, or the caller sizes it from mdqe_tracker_state first).
extern "C" int mdqe_tracker_get_result(void* handle, int is_last, float* bank_sum, float* bank_cnt, long hw, float* out_masks,
                                       float* carry, float* out_cls, int* n, int* ln, void* stream) {
  MDQE_CHECK_PTR(handle); MDQE_CHECK_PTR(bank_sum); MDQE_CHECK_PTR(bank_cnt); MDQE_CHECK_PTR(out_cls);
  MDQE_CHECK_PTR(n); MDQE_CHECK_PTR(ln);
  Tracker* t = static_cast<Tracker*>(handle);
  if (t->num_inst > 0) { MDQE_CHECK_PTR(out_masks); if (!is_last) MDQE_CHECK_PTR(carry); }   // (a window without any track: empty outputs)
  const int rc = t->result(is_last, out_cls, n, ln, nullptr);
  if (rc != MDQE_OK) return rc;
  int rc2 = mdqe_trk_window_mean_f32(bank_sum, bank_cnt, hw, t->mem_len, *n, *ln, 0, out_masks, stream);
  if (rc2 == MDQE_OK && !is_last) rc2 = mdqe_trk_carry_f32(bank_sum, bank_cnt, hw, t->mem_len, *n, t->mem_len - t->win, t->win, carry, stream);
  if (rc2 != MDQE_OK && !is_last) t->poisoned = true;                   // the host half has re-based its window, the device bank has not
  return rc2;
}
