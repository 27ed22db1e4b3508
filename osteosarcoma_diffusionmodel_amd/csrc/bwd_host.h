// bwd_host.h -- host side of the persistent backward launch (bwd_persist.h): the caller (train.hip) describes the dgrad GEMMs and
// the weight gradients of one backward pass in program order; launch() turns them into the two work queues, uploads them when
// the description changed since the previous step (it normally has not: same arena, same parameters) and starts the kernel.
#pragma once
#include <vector>
#include "gemm.h"
#include "kernels_train.h"

struct osd_handle;

namespace osd {

struct BwdDgradIn {
  int gw;                    // group width of the GroupNorm backward in the epilogue (32 | 64), 0 = plain store of dX
  int drop;                  // dropout sits behind that GroupNorm+SiLU
  GemmArgs g;
  GnBwdEpi e;                // gw == 0: e.gz / e.ldg = destination
  int dep0, dep1;            // tensor ids whose rows must be final (-1: none): the B operand, and the partial sum an accumulating epilogue reads
  int sig;                   // tensor id this GEMM's epilogue finalises (-1: none)
};
struct BwdWgradIn {
  WgPending w;
  int dep;                   // tensor id of its gz operand (-1: ready at launch)
};

class BwdBuilder {
 public:
  explicit BwdBuilder(int64_t rows) : rows_(rows) {}
  int new_tensor() { return n_tensors_++; }
  // false: the GEMM does not meet the preconditions of the persistent kernel's tile code (alignment, widths); the caller then
  // runs the whole pass on the per-launch path
  bool add_dgrad(const BwdDgradIn& d);
  bool add_wgrad(const WgPending& w, int dep);
  bool empty() const { return dg_.empty() && wg_.empty(); }
  // zeroes the counters, launches bwd_persist_kernel and the slab reduction of the split weight gradients on s
  int launch(osd_handle* h, hipStream_t s, float* slabs, int64_t slab_floats, uint64_t seed, uint32_t row_offset);

 private:
  int64_t rows_;
  int n_tensors_ = 0;
  std::vector<BwdDgradIn> dg_;
  std::vector<BwdWgradIn> wg_;
  friend struct BwdPlanDev;
};

void bwd_persist_free(osd_handle* h);
// OSD_EHIP if the last persistent backward launch gave up in a dependency wait (reads the status word: synchronises the stream)
int bwd_persist_check(osd_handle* h, hipStream_t s);

}  // namespace osd
