import torch, time
dev = torch.device("cuda:0")
N, C = 2449029, 128
def bench(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
for W, wn in ((8, 2), (8, 1), (4, 1), (2, 1)):
    wc = W // wn
    n_local = N // W; mr = n_local + 1
    Cw = C // wc; cs = Cw
    x = torch.randn(n_local, C, device=dev)
    send = torch.zeros(wn, wc, mr, cs, device=dev)
    def pack():
        send[:, :, :n_local] = x.view(n_local, wc, Cw)[:, :, 0:cs].transpose(0, 1).unsqueeze(0)
    recv = torch.randn(wc, mr, cs, device=dev)
    out = torch.empty(n_local, C, device=dev)
    def unpack():
        out.view(n_local, wc, Cw)[:, :, 0:cs] = recv[:, :n_local].transpose(0, 1)
    recv2 = torch.randn(wn, wc, mr, cs, device=dev)
    def unpack_sum():
        p = recv2[0] if wn == 1 else recv2.sum(0)
        out.view(n_local, wc, Cw)[:, :, 0:cs] = p[:, :n_local].transpose(0, 1)
    gsend = torch.zeros(wc, mr, cs, device=dev)
    def gpack():
        gsend[:, :n_local] = x.view(n_local, wc, Cw)[:, :, 0:cs].transpose(0, 1)
    mb = n_local * C * 4 / 1e6
    print(f"W={W} wn={wn} local {mb:.0f} MB: pack {bench(pack):.3f} ms, unpack {bench(unpack):.3f}, gpack {bench(gpack):.3f}, unpack_sum {bench(unpack_sum):.3f}")
