// CPU-only check of cvo::ShardPlan (include/UnifiedCvo/cvo/ShardPlan.hpp): the partition / gather arithmetic of
// cvo::CvoGPUSharded::align_batch driven without devices.  Every "device" fills its `per` send slots exactly as
// align_batch does (its pairs' ids, identity padding), the all-gather is a rank-major concatenation, and the read-back
// through slot_of() must return every pair's own record.  usage: cvo_shard_plan_check D n [n ...]
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "cvo/ShardPlan.hpp"

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const int D = std::atoi(argv[1]);
  for (int a = 2; a < argc; a++) {
    const int n = std::atoi(argv[a]);
    const cvo::ShardPlan plan(n, D);
    std::vector<int> gathered;  // what ncclAllGather leaves on every device: D blocks of `per` slots
    int covered = 0;
    for (int d = 0; d < D; d++) {
      std::vector<int> send(plan.per, -1);  // -1 = identity padding
      for (int q = 0; q < plan.count(d); q++) send[q] = plan.lo(d) + q;
      covered += plan.count(d);
      if (plan.lo(d) > plan.hi(d) || plan.count(d) > plan.per) return 1;
      if (d > 0 && plan.lo(d) != plan.hi(d - 1)) return 1;  // contiguous, no gap, no overlap
      gathered.insert(gathered.end(), send.begin(), send.end());
    }
    if (covered != n || (int)gathered.size() != plan.gathered_slots()) return 1;
    for (int p = 0; p < n; p++) {
      const int d = plan.device_of(p);
      if (d < 0 || d >= D || p < plan.lo(d) || p >= plan.hi(d)) return 1;
      if (gathered[plan.slot_of(p)] != p) return 1;
    }
    std::printf("D=%d n=%d per=%d counts=", D, n, plan.per);
    for (int d = 0; d < D; d++) std::printf("%d%s", plan.count(d), d + 1 < D ? "," : "\n");
  }
  return 0;
}
