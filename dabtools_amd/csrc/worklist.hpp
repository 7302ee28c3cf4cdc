// worklist.hpp — host side of the MSC decode: from the control plane's ETI jobs to the device work lists.
//
// No GPU call in here (the lists' allocator is a template parameter: the engine uses page-locked memory, the host-only sanitizer
// build of tests/host_sanitize plain std::allocator), so this file, control_plane.hpp, fifo_view.hpp and thread_pool.hpp are the
// units the CPU suite runs under ThreadSanitizer / AddressSanitizer / UBSan.
//
// What it restates of the reference: create_eti's loop over the active sub-channels in SubChId order (misc.c:241-281) -- which
// de-puncturing plan a sub-channel decodes with (uep_/eep_depuncture, depuncture.c:84-132), where its decoded bytes land in the
// frame (obytes = ((bits/8)+7) & 0xfff8, misc.c:259-260) -- turned inside out: all frames of all streams that share a code word
// shape are decoded together, 64 to a wave, longest code words first.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "control_plane.hpp"
#include "dab_tables.hpp"
#include "device_types.hpp"
#include "thread_pool.hpp"

namespace dabhip {

constexpr int kWorklistEtiBytes = 6144;

// Work list for gather + Viterbi launches: wave-groups of <= 64 equal-length code words.
template <template <class> class Alloc>
struct DecodeBatchT {
  std::vector<WaveGroup, Alloc<WaveGroup>> groups;   // longest code words first
  std::vector<int, Alloc<int>> job_ids;              // lanes of group g decode jobs job_ids[g.first .. g.first + g.count); padded to tiles of 64
  std::vector<int> slice_start;                      // launches: groups [slice_start[i], slice_start[i+1]) share the survivor-record buffer
  int64_t max_dec_rows = 0;
  bool wave_form = false;                            // few code words: decoded one WAVE per code word (k_vitwave.hip), the groups' dec_base laid out for it
};

// Everything the MSC decode of a set of ETI frames needs, prepared on the host (no GPU work)
template <template <class> class Alloc>
struct MscWorkT {
  std::vector<DecodeJob, Alloc<DecodeJob>> jobs;
  std::vector<EtiFrameMeta, Alloc<EtiFrameMeta>> meta;
  std::vector<uint8_t, Alloc<uint8_t>> headers;
  int header_stride = 0;
  DecodeBatchT<Alloc> batch;
  std::vector<int> stream_row_base;
  size_t nframes = 0;
};

inline CodewordPlan make_codeword_plan(const PuncturePlan& pp, int start_bit, int out_offset)
{
  CodewordPlan p;
  for (int s = 0; s < 4; ++s) { p.blocks[s] = pp.blocks[s]; p.mask[s] = puncture_mask(pp.pi[s]); }
  p.nsteps = pp.trellis_steps();
  p.start_bit = start_bit;
  p.out_offset = out_offset;
  p.out_bytes = (p.nsteps - 6) / 8;
  return p;
}

// The code word plans an engine has seen, identified by content (grow-only: ids stay valid, the table is uploaded whole)
class PlanTable {
 public:
  int id(const CodewordPlan& p)
  {
    std::vector<int32_t> key = {p.blocks[0], p.blocks[1], p.blocks[2], p.blocks[3],
                                static_cast<int32_t>(p.mask[0]), static_cast<int32_t>(p.mask[1]), static_cast<int32_t>(p.mask[2]),
                                static_cast<int32_t>(p.mask[3]), p.nsteps, p.start_bit, p.out_offset, p.out_bytes};
    auto it = index_.find(key);
    if (it != index_.end()) return it->second;
    const int id = static_cast<int>(plans_.size());
    plans_.push_back(p);
    index_.emplace(std::move(key), id);
    return id;
  }
  const std::vector<CodewordPlan>& plans() const { return plans_; }
  const CodewordPlan& operator[](int i) const { return plans_[static_cast<size_t>(i)]; }

 private:
  std::vector<CodewordPlan> plans_;
  std::map<std::vector<int32_t>, int> index_;
};

// wave-groups of <= 64 jobs per plan, longest code words first; plan_jobs[i] = (plan id, job indices decoded with that plan).
// (Measured in round 3 with an order knob: dealing the 3078- and 1542-step classes -- or the short ones -- proportionally into one
// another, or short words beside the longest, costs 6.1 .. 7.1 ms against 5.45 on the same box: longest-first stays.)
template <template <class> class Alloc>
void build_decode_batch(const PlanTable& plans, const std::vector<std::pair<int, const std::vector<int>*>>& plan_jobs, DecodeBatchT<Alloc>& out)
{
  out.groups.clear();
  out.job_ids.clear();
  out.slice_start.clear();
  out.max_dec_rows = 0;
  std::vector<std::pair<int, size_t>> order;   // (nsteps, index into plan_jobs)
  for (size_t i = 0; i < plan_jobs.size(); ++i) order.emplace_back(plans[plan_jobs[i].first].nsteps, i);
  std::stable_sort(order.begin(), order.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
  std::map<const std::vector<int>*, int> placed;   // plans of one layout share the same job list
  for (const auto& o : order) {
    const int plan = plan_jobs[o.second].first;
    const std::vector<int>& ids = *plan_jobs[o.second].second;
    auto it = placed.find(&ids);
    if (it == placed.end()) {
      out.job_ids.resize((out.job_ids.size() + 63) / 64 * 64, -1);        // lists start on a tile of 64 (regroup_kernel)
      it = placed.emplace(&ids, static_cast<int>(out.job_ids.size())).first;
      out.job_ids.insert(out.job_ids.end(), ids.begin(), ids.end());
    }
    const int first = it->second;
    for (size_t g = 0; g < ids.size(); g += 64)
      out.groups.push_back(WaveGroup{plan, first + static_cast<int>(g), static_cast<int>(std::min<size_t>(64, ids.size() - g)), o.first, 0, 0});
  }
}

// slices of wave-groups whose survivor records fit `max_rows` (rows of 64 x 8 bytes); record offsets are per slice.
// At most wave_max_codewords code words in all: the low-latency form instead (one wave per code word, k_vitwave.hip): one launch, every code
// word its own rows -- one per chunk of wave_chunk trellis steps.
template <template <class> class Alloc>
void plan_decode_batch(DecodeBatchT<Alloc>& b, int64_t max_rows, int64_t wave_max_codewords = 0, int wave_chunk = 60)
{
  b.slice_start = {0};
  b.max_dec_rows = 0;
  int64_t dec_rows = 0;
  const int ng = static_cast<int>(b.groups.size());
  int64_t codewords = 0;
  for (const WaveGroup& g : b.groups) codewords += g.count;
  b.wave_form = ng > 0 && codewords <= wave_max_codewords;
  if (b.wave_form) {
    for (int g = 0; g < ng; ++g) {
      b.groups[g].step_base = 0;
      b.groups[g].dec_base = dec_rows;
      dec_rows += int64_t(64) * ((b.groups[g].nsteps + wave_chunk - 1) / wave_chunk);
    }
    b.slice_start.push_back(ng);
    b.max_dec_rows = dec_rows;
    b.job_ids.resize((b.job_ids.size() + 63) / 64 * 64, -1);
    return;
  }
  for (int g = 0; g < ng; ++g) {
    const int64_t dr = (b.groups[g].nsteps + 7) / 8 * 8;
    if (g > b.slice_start.back() && dec_rows + dr > max_rows) {
      b.slice_start.push_back(g);
      b.max_dec_rows = std::max(b.max_dec_rows, dec_rows);
      dec_rows = 0;
    }
    b.groups[g].step_base = 0;
    b.groups[g].dec_base = dec_rows;
    dec_rows += dr;
  }
  b.slice_start.push_back(ng);
  b.max_dec_rows = std::max(b.max_dec_rows, dec_rows);
  b.job_ids.resize((b.job_ids.size() + 63) / 64 * 64, -1);     // tiles of 64 records
}

// The per-stream ETI jobs of a decode -> frame records (stream-major), ETI header rows and the decode batch.
// stream_row_base[b]: logical CIF row of stream b's CIF 0; stream_fib_base[b]: FIB block (4 per TF slot) of its CIF 0.
// Returns false (with *error set) when a multiplex does not fit an ETI frame.
template <template <class> class Alloc>
bool prepare_msc_work(PlanTable& plans, ThreadPool& pool, const std::vector<const JobList*>& stream_jobs,
                      const std::vector<const ControlPlane*>& planes, const std::vector<int>& stream_row_base,
                      const std::vector<int>& stream_fib_base, int64_t max_rows, MscWorkT<Alloc>& out, std::string* error,
                      const std::function<void(const char*)>& mark = nullptr, int64_t wave_max_codewords = 0)
{
  size_t nf = 0;
  for (const auto* v : stream_jobs) nf += v->size();
  out.nframes = nf;
  out.stream_row_base = stream_row_base;
  if (nf == 0) return true;
  // An ensemble layout (the active sub-channels in SubChId order) fixes the code word plans and their offsets in
  // the ETI frame.  Layouts are identified by content so that streams carrying the same multiplex share plans.
  struct Layout {
    std::vector<int> plan_ids;
    int mst_bytes = 0;
  };
  std::map<std::vector<int32_t>, int> layout_index;
  std::vector<Layout> layouts;
  std::vector<std::vector<int>> layout_frames;
  auto& jobs = out.jobs;
  auto& meta = out.meta;
  const size_t nstreams = stream_jobs.size();
  // pass 0 (parallel over streams; the job lists are ~15 MB, walked once here): the longest header and the layouts a stream uses
  // (layouts change rarely: one entry per run), with the header length of the first job of each
  std::vector<int> stream_max_header(nstreams, 0);
  std::vector<std::vector<std::pair<int, int>>> used(nstreams);   // (local layout, header_len)
  pool.parallel_for(static_cast<int>(nstreams), [&](int b) {
    int mh = 0, prev = -1;
    for (const EtiJob& j : *stream_jobs[b]) {
      mh = std::max(mh, j.header_len);
      if (j.layout != prev) {
        prev = j.layout;
        bool seen = false;
        for (const auto& u : used[b]) seen = seen || u.first == j.layout;
        if (!seen) used[b].push_back({j.layout, j.header_len});
      }
    }
    stream_max_header[b] = mh;
  });
  int max_header = 0;
  for (int mh : stream_max_header) max_header = std::max(max_header, mh);
  const int header_stride = (max_header + 15) & ~15;
  out.header_stride = header_stride;
  auto& headers = out.headers;
  // every record is written in full by pass 2 (a header row up to its own length, which is all K5 reads): no fill
  jobs.resize(nf);
  meta.resize(nf);
  headers.resize(nf * static_cast<size_t>(header_stride));
  if (mark) mark("lists sized");
  // pass 1 (serial, cheap): global layout id of every (stream, local layout)
  std::vector<std::vector<int>> local_to_global(nstreams);
  std::vector<size_t> frame_base(nstreams + 1, 0);
  for (size_t b = 0; b < nstreams; ++b) {
    frame_base[b + 1] = frame_base[b] + stream_jobs[b]->size();
    if (stream_jobs[b]->empty()) continue;
    const auto& lays = planes[b]->layouts();
    local_to_global[b].assign(lays.size(), -1);
    for (const auto& u : used[b]) {
      const int layout = u.first, job_header_len = u.second;
      const std::vector<SubChannel>& subs = lays[layout];
      std::vector<int32_t> key = {job_header_len};
      for (const SubChannel& sc : subs) {
        const int32_t fields[] = {sc.slform, sc.uep_index, sc.start_cu, sc.size_cu, sc.bitrate, sc.protlev};
        key.insert(key.end(), fields, fields + 6);
      }
      auto it = layout_index.find(key);
      if (it == layout_index.end()) {
        Layout lay;
        int off = job_header_len + 96;
        for (const SubChannel& sc : subs) {
          CodewordPlan cp = make_codeword_plan(puncture_plan(sc), sc.start_cu * 64, off);
          lay.plan_ids.push_back(plans.id(cp));
          off += (cp.out_bytes + 7) & 0xfff8;          // misc.c:259-260: obytes = ((bits/8)+7) & 0xfff8
        }
        lay.mst_bytes = off - job_header_len - 96;
        if (off + 8 > kWorklistEtiBytes) {
          if (error) *error = "ETI frame overflow: sub-channels exceed 6144 bytes";
          return false;
        }
        it = layout_index.emplace(std::move(key), static_cast<int>(layouts.size())).first;
        layouts.push_back(std::move(lay));
        layout_frames.emplace_back();
      }
      local_to_global[b][layout] = it->second;
    }
  }
  if (mark) mark("layouts");
  // pass 2 (parallel over streams): per-frame records
  std::vector<std::vector<std::pair<int, std::pair<int, int>>>> runs(nstreams);   // per stream: (layout, [first, last) frame)
  pool.parallel_for(static_cast<int>(nstreams), [&](int b) {
    size_t f = frame_base[b];
    int run_gid = -1;
    for (const EtiJob& job : *stream_jobs[b]) {
      const int gid = local_to_global[b][job.layout];
      if (gid != run_gid) {
        runs[b].push_back({gid, {static_cast<int>(f), static_cast<int>(f)}});
        run_gid = gid;
      }
      runs[b].back().second.second = static_cast<int>(f) + 1;
      jobs[f] = DecodeJob{static_cast<int32_t>(b), job.first_cif};
      meta[f] = EtiFrameMeta{job.header_len, layouts[gid].mst_bytes, stream_fib_base[b] + job.first_cif, 0};
      std::memcpy(headers.data() + f * header_stride, stream_jobs[b]->header(job), static_cast<size_t>(job.header_len));
      ++f;
    }
  });
  if (mark) mark("frame records");
  for (size_t b = 0; b < nstreams; ++b)
    for (const auto& r : runs[b])
      for (int f = r.second.first; f < r.second.second; ++f) layout_frames[r.first].push_back(f);
  if (mark) mark("layout frames");
  std::vector<std::pair<int, const std::vector<int>*>> plan_jobs;
  for (size_t l = 0; l < layouts.size(); ++l)
    for (int pid : layouts[l].plan_ids) plan_jobs.emplace_back(pid, &layout_frames[l]);
  build_decode_batch(plans, plan_jobs, out.batch);
  plan_decode_batch(out.batch, max_rows, wave_max_codewords);
  if (mark) mark("batch");
  return true;
}

}  // namespace dabhip
