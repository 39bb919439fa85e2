// Host-side graph constants of the GAT encoder (no GPU).  C++ replacement for the Cython algos.pyx the
// reference depends on but does not ship (lib/models/backbones/setup.py:1-6; outputs normally arrive as
// data/base_data/{shortest_path,path}_{h36m,3dpw}.npy, lib/models/GAT.py:89-93), plus gen_edg_input
// (lib/models/backbones/modules.py:6-29) and build_verts_joints_relation (lib/graph_utils.py:71-89).
#include <cstdint>
#include <vector>

#include "gator_hip.h"
#include "internal.h"

namespace {
constexpr int64_t kSentinel = 510;  // unreachable / "no intermediate node", modules.py:8,22

// Iterative expansion of the intermediate-node matrix: nodes strictly between i and j, in order.
void expand_path(const int64_t* path, int n, int i, int j, std::vector<int>& out) {
    const int64_t k = path[(int64_t)i * n + j];
    if (k == kSentinel) return;
    expand_path(path, n, i, (int)k, out);
    out.push_back((int)k);
    expand_path(path, n, (int)k, j, out);
}
}  // namespace

extern "C" int gator_floyd_warshall(const float* adj, int32_t n, int64_t* dist, int64_t* path) {
    if (!adj || !dist || !path || n <= 0 || n > 64) return gator::fail(GATOR_EINVAL, "gator_floyd_warshall: bad arguments");
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            dist[i * n + j] = (i == j) ? 0 : (adj[i * n + j] != 0.f ? 1 : kSentinel);
            path[i * n + j] = kSentinel;
        }
    for (int k = 0; k < n; ++k)
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                const int64_t c = dist[i * n + k] + dist[k * n + j];
                if (dist[i * n + j] > c) {
                    dist[i * n + j] = c;
                    path[i * n + j] = k;
                }
            }
    for (int i = 0; i < n * n; ++i)
        if (dist[i] >= kSentinel) {
            dist[i] = kSentinel;
            path[i] = kSentinel;
        }
    return GATOR_OK;
}

extern "C" int gator_gen_edge_input(const int64_t* path, const float* edge_len, int32_t n, int32_t max_dist, float* out) {
    if (!path || !edge_len || !out || n <= 0 || n > 64 || max_dist <= 0) return gator::fail(GATOR_EINVAL, "gator_gen_edge_input: bad arguments");
    for (int64_t i = 0; i < (int64_t)n * n * max_dist; ++i) out[i] = 0.f;
    std::vector<int> p;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            if (i == j || path[i * n + j] == kSentinel) continue;   // modules.py:20-23
            p.clear();
            p.push_back(i);
            expand_path(path, n, i, j, p);
            p.push_back(j);
            if ((int)p.size() - 1 > max_dist) return gator::fail(GATOR_ESHAPE, "gator_gen_edge_input: path longer than max_dist");
            for (size_t k = 0; k + 1 < p.size(); ++k)
                out[((int64_t)i * n + j) * max_dist + k] = edge_len[p[k] * n + p[k + 1]];
        }
    return GATOR_OK;
}

extern "C" int gator_verts_joints_relation(const float* joints, int32_t n_joint, const float* verts, int32_t n_vert, int32_t* rel) {
    if (!joints || !verts || !rel || n_joint <= 0 || n_vert <= 0) return gator::fail(GATOR_EINVAL, "gator_verts_joints_relation: bad arguments");
    for (int v = 0; v < n_vert; ++v) {
        int best = 0;
        float bd = 0.f;
        for (int j = 0; j < n_joint; ++j) {
            // float32 arithmetic in the reference's order: (v - joints)**2 summed over xyz (graph_utils.py:81-84)
            float d = 0.f;
            for (int c = 0; c < 3; ++c) {
                const float t = verts[v * 3 + c] - joints[j * 3 + c];
                d += t * t;
            }
            if (j == 0 || d < bd) {   // np.argmin: first minimum
                bd = d;
                best = j;
            }
        }
        rel[v] = best;
    }
    return GATOR_OK;
}
