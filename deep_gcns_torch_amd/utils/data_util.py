"""OGB feature-table sizes and the random node partition helpers the sparse library and the
OGB example scripts import from `utils.data_util` (reference: utils/data_util.py:14-61,246-348).
No h5py / torch_scatter / PyG imports at module load."""
import numpy as np
import torch

# sizes of the OGB categorical feature vocabularies (ogb/utils/features.py; reference
# utils/data_util.py:246-282 lists the same tables, each ends with a 'misc' bucket)
_ATOM_DIMS = [119, 4, 12, 12, 10, 6, 6, 2, 2]   # atomic num, chirality, degree, charge, numH, radical e, hybridisation, aromatic, in ring
_BOND_DIMS = [5, 6, 2]                            # bond type, stereo, conjugated


def get_atom_feature_dims():
    return list(_ATOM_DIMS)


def get_bond_feature_dims():
    return list(_BOND_DIMS)


def intersection(lst1, lst2):
    return list(set(lst1) & set(lst2))


def process_indexes(idx_list):
    """Map each id to its position and return the ids ordered by id (utils/data_util.py:18-23)."""
    pos = {idx: i for i, idx in enumerate(idx_list)}
    return [pos[k] for k in sorted(pos)]


def random_partition_graph(num_nodes, cluster_number=10):
    """Uniform random cluster id per node (utils/data_util.py:43-45)."""
    return np.random.randint(cluster_number, size=num_nodes)


def generate_sub_graphs(adj, parts, cluster_number=10, batch_size=1):
    """Induced sub-graph of every cluster (utils/data_util.py:48-61): returns (node id arrays, COO edge_index
    tensors).  ``adj`` is what the reference's callers pass, a ``torch_sparse.SparseTensor`` (anything with
    ``to_scipy(layout='csr')``), or directly a scipy sparse matrix.  As in the reference, batch ``c`` holds the
    nodes with ``parts == c`` for c in range(cluster_number // batch_size) -- ``batch_size`` only changes how
    many batches are produced.  Edge order = scipy's ``tocoo()`` of the sliced CSR (row-major), which is what
    ``torch_geometric.utils.from_scipy_sparse_matrix`` returns.  For the device-side equivalent on an
    ``edge_index`` see ``deep_gcns_torch_amd.graph_prep.induced_subgraph``."""
    if hasattr(adj, "to_scipy"):
        adj = adj.to_scipy(layout='csr')
    else:
        adj = adj.tocsr()
    num_batches = cluster_number // batch_size
    sg_nodes, sg_edges = [], []
    for cluster in range(num_batches):
        nodes = np.where(parts == cluster)[0]
        coo = adj[nodes, :][:, nodes].tocoo()
        sg_nodes.append(nodes)
        sg_edges.append(torch.from_numpy(np.vstack((coo.row, coo.col))).long())
    return sg_nodes, sg_edges


# ---- dataset-preparation helpers the OGB graph-property examples import (data prep on CPU tensors, not
# ---- part of the message-passing hot path; reference utils/data_util.py:26-40) --------------------------
def add_zeros(data):
    data.x = torch.zeros(data.num_nodes, dtype=torch.long)
    return data


def extract_node_feature(data, reduce='add'):
    """Node features = reduction of the incident edge features by source node."""
    if reduce not in ['mean', 'max', 'add']:
        raise Exception('Unknown Aggregation Type')
    idx = data.edge_index[0]
    out = torch.zeros((data.num_nodes,) + tuple(data.edge_attr.shape[1:]), dtype=data.edge_attr.dtype,
                      device=data.edge_attr.device)
    index = idx.view(-1, *([1] * (data.edge_attr.dim() - 1))).expand_as(data.edge_attr)
    kind = {'add': 'sum', 'mean': 'mean', 'max': 'amax'}[reduce]
    data.x = out.scatter_reduce(0, index, data.edge_attr, kind, include_self=False)
    return data


def __getattr__(name):
    if name == "PartNet":
        raise ImportError("utils.data_util.PartNet is a torch_geometric InMemoryDataset (dataset code, out of the "
                          "hot-path scope); import it from the reference's utils with torch_geometric installed")
    raise AttributeError(name)
