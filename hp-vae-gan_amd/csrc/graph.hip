// Census of a captured hipGraph's nodes (host code only).
//
// Why it exists: on this runtime (ROCm 7.2) hipMemsetAsync / hipMemcpyAsync NODES inside a captured
// hipGraph are not reliably ordered against the kernel nodes around them - a train iteration replayed from such a graph
// silently trained NaNs.  The library itself launches kernels only (tests/test_host_cpu.py scans the sources), but the
// host side of a captured iteration also runs torch ops (fills, slices, pads), and how torch lowers those is not ours to
// decide: StageTrainer.enable_graph() counts the node types of what was actually captured and refuses a graph that
// holds anything but kernel (and empty / event) nodes.
#include "hpvg_common.h"
#include "hpvg.h"
#include <vector>

extern "C" {

// counts[t] = number of nodes of hipGraphNodeType t (t < ntypes; nodes of larger type ids are added to counts[ntypes-1]);
// child graphs are descended into.  graph: a hipGraph_t (torch.cuda.CUDAGraph(keep_graph=True).raw_cuda_graph()).
int hpvg_graph_node_census(void* graph, int* counts, int ntypes) {
  if (!graph || !counts || ntypes < 1) return HPVG_ERR_ARG;
  for (int i = 0; i < ntypes; ++i) counts[i] = 0;
  std::vector<hipGraph_t> todo{(hipGraph_t)graph};
  while (!todo.empty()) {
    hipGraph_t g = todo.back();
    todo.pop_back();
    size_t n = 0;
    if (hipGraphGetNodes(g, nullptr, &n) != hipSuccess) return HPVG_ERR_LAUNCH;
    std::vector<hipGraphNode_t> nodes(n);
    if (n && hipGraphGetNodes(g, nodes.data(), &n) != hipSuccess) return HPVG_ERR_LAUNCH;
    for (size_t i = 0; i < n; ++i) {
      hipGraphNodeType t;
      if (hipGraphNodeGetType(nodes[i], &t) != hipSuccess) return HPVG_ERR_LAUNCH;
      const int k = (int)t < ntypes ? (int)t : ntypes - 1;
      counts[k] += 1;
      if (t == hipGraphNodeTypeGraph) {
        hipGraph_t child = nullptr;
        if (hipGraphChildGraphNodeGetGraph(nodes[i], &child) == hipSuccess && child) todo.push_back(child);
      }
    }
  }
  return HPVG_OK;
}

}  // extern "C"
