"""attention_rows_kernel: run-to-run determinism, independence of a row's result from the number of rows in the launch, equality with the
per-query kernel — on shapes like the encoder's (E = 320, 4 heads) with ragged klen.  Diagnostic for the world-size test."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manga_image_translator_amd import lib as L, ops
lib = L.load()
dev = torch.device("cuda:0")
st = C.c_void_p(ops.current_stream())
heads, hd = 4, 80
E = heads * hd
g = torch.Generator().manual_seed(11)
def run(q, k, v, klen, R, T):
    out = torch.full((R, T, E), float("nan"), device=dev)
    L.check(lib.mit_attention_heads(q.data_ptr(), T * E, E, k.data_ptr(), T * E, E, v.data_ptr(), T * E, E, out.data_ptr(), T * E, E,
                                    klen.data_ptr(), R, T, T, 1, heads, hd, st), "rows")
    return out
bad = 0
for T in (24, 47, 64, 97, 113, 142):
    R = 12
    q, k, v = (torch.randn(R, T, E, generator=g).to(dev) for _ in range(3))
    klen = torch.randint(1, T + 1, (R,), generator=g).to(torch.int32).to(dev)
    a = run(q, k, v, klen, R, T)
    for rep in range(20):
        b = run(q, k, v, klen, R, T)
        if not torch.equal(a, b):
            d = (a != b) & ~(a.isnan() & b.isnan())
            idx = d.nonzero()
            print(f"T={T} rep {rep}: run-to-run {int(d.sum())} values differ, first {idx[:4].tolist()}"); bad += 1; break
    h = run(q[:6].contiguous(), k[:6].contiguous(), v[:6].contiguous(), klen[:6].contiguous(), 6, T)
    if not torch.equal(a[:6], h):
        d = (a[:6] != h)
        print(f"T={T}: rows 0..5 differ between R=12 and R=6 launches: {int(d.sum())} values, first {d.nonzero()[:6].tolist()}, klen {klen[:6].tolist()}"); bad += 1
    per = torch.full_like(a, float("nan"))
    for t in range(T):
        L.check(lib.mit_attention_heads(q[:, t].data_ptr(), T * E, E, k.data_ptr(), T * E, E, v.data_ptr(), T * E, E, per[:, t].data_ptr(),
                                        T * E, E, klen.data_ptr(), R, 1, T, 1, heads, hd, st), "per-query")
    if not torch.equal(a, per):
        d = (a != per)
        print(f"T={T}: differs from the per-query kernel: {int(d.sum())} values, first {d.nonzero()[:6].tolist()}, klen {klen.tolist()}"); bad += 1
    # a longer padded length with the same valid lengths must not change the valid outputs (queries < T only)
    T2 = T + 19
    q2, k2, v2 = (torch.zeros(R, T2, E, device=dev) for _ in range(3))
    q2[:, :T], k2[:, :T], v2[:, :T] = q, k, v
    k2[:, T:] = 7.0; v2[:, T:] = -3.0
    c = run(q2, k2, v2, klen, R, T2)
    if not torch.equal(c[:, :T], a):
        d = (c[:, :T] != a)
        print(f"T={T}: padded length {T2} changes {int(d.sum())} values, first {d.nonzero()[:6].tolist()}"); bad += 1
print("DIAG", "FAILED" if bad else "OK", bad)
