"""Oracle (test infrastructure): load-time graph constants of the GAT encoder.

Plain numpy / pure-Python restatement of the reference's constant construction.
Every function cites the reference file:line it follows (paths relative to
/root/reference).  Small J (17 / 19) so Python loops are fine.
"""
import math

import numpy as np

SENTINEL = 510  # "unreachable / no intermediate node" marker, lib/models/backbones/modules.py:8,22

# Joint-set tables (data).  data/Human36M/dataset.py:56-59 (h36m), :70-74 (coco);
# the same tables appear in data/PW3D/dataset.py:57-61 and demo/run.py:72-87.
H36M_SKELETON = ((0, 7), (7, 8), (8, 9), (9, 10), (8, 11), (11, 12), (12, 13), (8, 14), (14, 15),
                 (15, 16), (0, 1), (1, 2), (2, 3), (0, 4), (4, 5), (5, 6))
H36M_FLIP_PAIRS = ((1, 4), (2, 5), (3, 6), (14, 11), (15, 12), (16, 13))
COCO_SKELETON = ((1, 2), (0, 1), (0, 2), (2, 4), (1, 3), (6, 8), (8, 10), (5, 7), (7, 9), (12, 14),
                 (14, 16), (11, 13), (13, 15), (17, 11), (17, 12), (17, 18), (18, 5), (18, 6), (18, 0))
COCO_FLIP_PAIRS = ((1, 2), (3, 4), (5, 6), (7, 8), (9, 10), (11, 12), (13, 14), (15, 16))


def joint_setting(num_joint):
    if num_joint == 17:
        return H36M_SKELETON, H36M_FLIP_PAIRS
    if num_joint == 19:
        return COCO_SKELETON, COCO_FLIP_PAIRS
    raise ValueError("reference supports J=17 (h36m) or J=19 (coco+pelvis+neck) only, lib/models/GAT.py:79-93")


def build_adj(joint_num, skeleton, flip_pairs):
    """lib/graph_utils.py:60-69 -- skeleton + flip pairs + identity, float64."""
    adj = np.zeros((joint_num, joint_num))
    for a, b in skeleton:
        adj[a, b] = 1
        adj[b, a] = 1
    for a, b in flip_pairs:
        adj[a, b] = 1
        adj[b, a] = 1
    return adj + np.eye(joint_num)


def delete_symmetric_edges(adj):
    """lib/models/GAT.py:58-65 -- hard-coded h36m indices applied to every joint set; float32 dense."""
    a = np.array(adj, dtype=np.float32, copy=True)
    for i, j in ((1, 4), (2, 5), (3, 6), (11, 14), (12, 15), (13, 16)):
        a[i, j] = 0
        a[j, i] = 0
    return a


def floyd_warshall(adj):
    """Graphormer ``algos.pyx`` (absent from the reference tree; lib/models/backbones/setup.py:1-6 is its
    build stub) with GATOR's 510 convention inferred from lib/models/backbones/modules.py:8,22:
    distance 510 = unreachable, path[i][j] = 510 = no intermediate node.  Returns (dist, path) int64."""
    n = adj.shape[0]
    m = (np.asarray(adj) != 0).astype(np.int64)
    path = np.full((n, n), SENTINEL, dtype=np.int64)
    for i in range(n):
        for j in range(n):
            if i == j:
                m[i, j] = 0
            elif m[i, j] == 0:
                m[i, j] = SENTINEL
    for k in range(n):
        for i in range(n):
            for j in range(n):
                c = m[i, k] + m[k, j]
                if m[i, j] > c:
                    m[i, j] = c
                    path[i, j] = k
    for i in range(n):
        for j in range(n):
            if m[i, j] >= SENTINEL:
                path[i, j] = SENTINEL
                m[i, j] = SENTINEL
    return m, path


def get_all_edges(path, i, j):
    """lib/models/backbones/modules.py:6-11."""
    k = int(path[i][j])
    if k == SENTINEL:
        return []
    return get_all_edges(path, i, k) + [k] + get_all_edges(path, k, j)


def gen_edg_input(max_dist, path, edge_feat):
    """lib/models/backbones/modules.py:13-29 -- [n, n, max_dist] float32 edge lengths along expand(path)."""
    n = path.shape[0]
    out = np.zeros((n, n, int(max_dist)), dtype=np.float32)
    for i in range(n):
        for j in range(n):
            if i == j:
                continue
            if path[i][j] == SENTINEL:
                continue
            p = [i] + get_all_edges(path, i, j) + [j]
            for k in range(len(p) - 1):
                out[i, j, k] = edge_feat[p[k], p[k + 1]]
    return out


def template_joints(j_regressor, mean_vertices, num_joint):
    """lib/models/GAT.py:74-88 -- float32 matmul, then pelvis/neck appended for the 19-joint set."""
    tj = (np.asarray(j_regressor, np.float32) @ np.asarray(mean_vertices, np.float32)).astype(np.float32)
    if num_joint == 19:
        pelvis = (tj[11] + tj[12]) * np.float32(0.5)
        neck = (tj[5] + tj[6]) * np.float32(0.5)
        tj = np.concatenate([tj, pelvis[None], neck[None]], 0)
    return tj


def edge_length_table(graph_adj, tj):
    """lib/models/GAT.py:95-108 -- UPPER-TRIANGULAR table of template edge lengths (float32 storage)."""
    n = graph_adj.shape[0]
    ed = np.zeros((n, n), dtype=np.float32)
    for i in range(n):
        for j in range(i + 1, n):
            if graph_adj[i][j] == 1:
                d = tj[i].astype(np.float32) - tj[j].astype(np.float32)
                ed[i, j] = math.sqrt(float((d * d).sum(dtype=np.float32)))
    return ed


def build_verts_joints_relation(joints, vertices):
    """lib/graph_utils.py:71-89 -- nearest template joint per coarse vertex (float64 index array there)."""
    rel = np.zeros((vertices.shape[0],), dtype=np.int64)
    for idx, v in enumerate(vertices):
        d = ((v - joints) ** 2).sum(1)
        rel[idx] = int(np.argmin(d))
    return rel


def downsample(mean_vertices, d_mats):
    """lib/models/backbones/mesh.py:93-108 via lib/models/MDR.py:79-81: 6890 -> 1723 -> 431 (float32 spmm)."""
    x = np.asarray(mean_vertices, np.float32)
    for d in d_mats:
        x = np.asarray(d.astype(np.float32) @ x, np.float32)
    return x


def gat_constants(num_joint, j_regressor, mean_vertices, shortest_path=None, path=None):
    """Everything GAT.__init__ derives (lib/models/GAT.py:56-112)."""
    skeleton, flips = joint_setting(num_joint)
    graph_adj = delete_symmetric_edges(build_adj(num_joint, skeleton, flips))
    if shortest_path is None or path is None:
        shortest_path, path = floyd_warshall(graph_adj)
    tj = template_joints(j_regressor, mean_vertices, num_joint)
    ed = edge_length_table(graph_adj, tj)
    max_dist = int(np.amax(shortest_path))
    edge_input = gen_edg_input(max_dist, path, ed)
    return dict(graph_adj=graph_adj, shortest_path=np.asarray(shortest_path, np.int64),
                path=np.asarray(path, np.int64), template_joints=tj, edge_len=ed,
                edge_input=edge_input, max_dist=max_dist)
