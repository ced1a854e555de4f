#!/usr/bin/env python
"""Read the candidate lists of the 32-row kNN filter kernel back from the workspace and check every (key, id) entry against
the oracle's distances (round 6: how the nondeterministic exclude_self fault of the C = 32 instantiation was found -- 16
entries of one candidate tile with |x_i|^2 missing from their distance, always rows 13 / 29 of a workgroup).

    python benchmarks/knn_list_readback.py          (GPU box; calls dgcn_knn_dense_f32 directly with its own workspace)
"""
import sys, torch, struct
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep_gcns_torch_amd import dense_ops, synth, _lib
from oracle import dense_ref
dev = torch.device("cuda:0")
lib = _lib.load()
N, C, K, d, ex = 2048, 32, 48, 3, True
B = 2
x = synth.lattice_cloud(B, C, N, seed=N + K)
dist = dense_ref.pairwise_distance(x.transpose(2, 1).squeeze(-1))
dist.diagonal(dim1=1, dim2=2).fill_(float("inf"))
srt = torch.sort(dist, dim=2).values
want = srt[:, :, :K:d]
x3 = x.squeeze(-1).to(dev)
ws_bytes = lib.dgcn_knn_dense_workspace_bytes(B, N, C)
pts = B * N
head = (pts * 12 + 4 + 255) // 256 * 256
planes = B * ((N + 15) // 16 * 16) * C * 6
def key_to_float(k):
    k = k & 0xFFFFFFFF
    b = (k & 0x7FFFFFFF) if (k & 0x80000000) else (~k & 0xFFFFFFFF)
    return struct.unpack("f", struct.pack("I", b))[0]
for rep in range(6):
    ws = torch.zeros(ws_bytes, device=dev, dtype=torch.uint8)
    nn = torch.empty(B, N, K // d, dtype=torch.int64, device=dev)
    ctr = torch.empty_like(nn)
    rc = lib.dgcn_knn_dense_f32(x3.data_ptr(), x3.stride(0), x3.stride(1), x3.stride(2), B, C, N, K, d, 1, nn.data_ptr(), ctr.data_ptr(), ws.data_ptr(), ws_bytes, _lib.current_stream_handle(dev))
    torch.cuda.synchronize()
    assert rc == 0
    g = nn.cpu()
    bad = torch.nonzero((torch.gather(dist, 2, g) != want).any(2)).tolist()
    lists = ws[head + planes: head + planes + pts * 1024 * 8].view(torch.int32).view(pts, 1024, 2).cpu()
    cnts = ws[head + planes + pts * 1024 * 8: head + planes + pts * 1024 * 8 + pts * 4].view(torch.int32).cpu()
    print("rep", rep, "bad rows", bad[:6])
    for b, i in bad[:3]:
        row = b * N + i
        c = int(cnts[row])
        ent = lists[row, :c]
        ids = ent[:, 1].long()
        keys = [key_to_float(int(k)) for k in ent[:, 0].tolist()]
        true = dist[b, i, ids.clamp(0, N - 1)]
        wrong = [(j, int(ids[j]), round(keys[j], 4), round(float(true[j]), 4)) for j in range(c) if not (0 <= int(ids[j]) < N) or abs(keys[j] - float(true[j])) > 1e-6]
        dup = c - len(set(ids.tolist()))
        print("   row", (b, i), "count", c, "duplicate ids", dup, "entries whose key != true distance (pos, id, key, true):", wrong[:8], "n_wrong", len(wrong))
