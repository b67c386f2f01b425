// Launch batcher: see batch.h.  Host code only.
#include "batch.h"

#include <cstring>
#include <vector>

#include "common.h"
#include "dav_kernels.h"

namespace davb {
namespace {

struct Op {
  GroupFn fn = nullptr;                 // typed (groupable) launch ...
  std::vector<unsigned char> blob;      // ... and its parameter block
  hipStream_t stream = nullptr;
  std::function<void()> opaque;         // or: anything else, replayed as recorded
};

typedef std::vector<Op> Step;            // launches of ONE lane that are independent of each other (usually just one)

struct State {
  bool on = false;
  bool auto_lanes = false;              // every recorded launch is its own lane (the whole batch is one region)
  bool in_region = false;               // launches recorded now join the current step of the current lane
  bool region_open = false;             // ... which has been started
  bool suspended = false;               // launches go out at once although a batch is open (weight-cache casts)
  std::vector<std::vector<Step>> lanes;
  int last_ops = 0, last_launches = 0;
};
thread_local State S;

Step& cur_step() {
  if (S.lanes.empty() || S.auto_lanes) S.lanes.emplace_back();
  auto& lane = S.lanes.back();
  if (!(S.in_region && S.region_open)) {
    lane.emplace_back();
    S.region_open = S.in_region;
  }
  return lane.back();
}

}  // namespace

bool recording() { return S.on && !S.suspended; }

void push_opaque(std::function<void()> fn) {
  Op op;
  op.opaque = std::move(fn);
  cur_step().push_back(std::move(op));
}

void push_typed(GroupFn fn, const void* params, size_t bytes, hipStream_t stream) {
  Op op;
  op.fn = fn;
  op.blob.assign((const unsigned char*)params, (const unsigned char*)params + bytes);
  op.stream = stream;
  cur_step().push_back(std::move(op));
}

}  // namespace davb

using namespace davb;

extern "C" int dav_batch_begin(int auto_lanes) {
  if (S.on) return DAV_ERR_SHAPE;       // no nesting
  S.on = true;
  S.auto_lanes = auto_lanes != 0;
  S.in_region = S.region_open = S.suspended = false;
  S.lanes.clear();
  return DAV_OK;
}

extern "C" int dav_batch_region(int begin) {
  if (!S.on) return DAV_ERR_SHAPE;
  if (begin && S.in_region) return DAV_ERR_SHAPE;      // regions do not nest
  S.in_region = begin != 0;
  S.region_open = false;
  return DAV_OK;
}

extern "C" int dav_batch_skip(int steps) {
  if (!S.on || S.auto_lanes || S.in_region || steps < 0) return DAV_ERR_SHAPE;
  if (S.lanes.empty()) S.lanes.emplace_back();
  for (int i = 0; i < steps; ++i) S.lanes.back().emplace_back();      // empty steps: this lane idles while the others advance
  return DAV_OK;
}

extern "C" int dav_batch_suspend(int on) {
  S.suspended = on != 0;
  return DAV_OK;
}

extern "C" int dav_batch_lane(void) {
  if (!S.on) return DAV_ERR_SHAPE;
  if (!S.auto_lanes && (S.lanes.empty() || !S.lanes.back().empty())) S.lanes.emplace_back();
  S.in_region = S.region_open = false;
  return DAV_OK;
}

extern "C" int dav_batch_end(void) {
  if (!S.on) return DAV_ERR_SHAPE;
  S.on = false;                          // from here on launches are real
  size_t steps = 0;
  int ops = 0, launches = 0;
  for (auto& l : S.lanes) steps = l.size() > steps ? l.size() : steps;
  std::vector<const void*> params;
  std::vector<char> done;
  std::vector<Op*> row;
  for (size_t k = 0; k < steps; ++k) {
    // the k-th steps of all lanes are mutually independent: bucket their typed launches by (family+configuration, stream)
    row.clear();
    for (auto& l : S.lanes)
      if (k < l.size())
        for (auto& op : l[k]) row.push_back(&op);
    ops += (int)row.size();
    done.assign(row.size(), 0);
    for (size_t i = 0; i < row.size(); ++i) {
      if (done[i]) continue;
      Op* a = row[i];
      if (!a->fn) {
        a->opaque();
        ++launches;
        continue;
      }
      params.clear();
      for (size_t j = i; j < row.size(); ++j) {
        Op* b = row[j];
        if (!done[j] && b->fn == a->fn && b->stream == a->stream) {
          params.push_back(b->blob.data());
          done[j] = 1;
        }
      }
      a->fn(params.data(), (int)params.size(), a->stream);
      ++launches;
    }
  }
  S.lanes.clear();
  S.last_ops = ops;
  S.last_launches = launches;
  return dav_launch_status();
}

extern "C" int dav_batch_abort(void) {
  S.on = S.in_region = S.region_open = false;
  S.lanes.clear();
  return DAV_OK;
}

extern "C" int dav_batch_stats(int* recorded_ops, int* issued_launches) {
  if (recorded_ops) *recorded_ops = S.last_ops;
  if (issued_launches) *issued_launches = S.last_launches;
  return DAV_OK;
}
