import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unboundednerfpytorch_amd import _lib
L = _lib.load()
torch.manual_seed(0)
for M in (5000, 20000, 40000, 70000):
    K = 39
    feat = torch.randn(M, K, device="cuda"); go = torch.randn(M, 3, device="cuda")
    w0 = torch.randn(128, K, device="cuda") * 0.1; b0 = torch.randn(128, device="cuda") * 0.1
    w1 = torch.randn(128, 128, device="cuda") * 0.1; b1 = torch.randn(128, device="cuda") * 0.1
    w2 = torch.randn(3, 128, device="cuda") * 0.1; b2 = torch.randn(3, device="cuda") * 0.1
    h1 = torch.empty(M, 128, device="cuda"); h2 = torch.empty(M, 128, device="cuda"); lg = torch.empty(M, 3, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(L.ugrid_rgbnet_train_forward(feat.data_ptr(), M, K, w0.data_ptr(), b0.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), 128, h1.data_ptr(), h2.data_ptr(), lg.data_ptr(), st), "f")
    rh1 = torch.relu(feat.double() @ w0.double().t() + b0.double()); rh2 = torch.relu(rh1 @ w1.double().t() + b1.double()); rl = rh2 @ w2.double().t() + b2.double()
    print(M, "fwd", float((h1 - rh1).abs().max()), float((h2 - rh2).abs().max()), float((lg - rl).abs().max()))
    gk = torch.empty(M, 12, device="cuda")
    g = [torch.empty_like(w0), torch.empty(128, device="cuda"), torch.empty_like(w1), torch.empty(128, device="cuda"), torch.empty_like(w2), torch.empty(3, device="cuda")]
    sc = torch.empty(int(L.ugrid_rgbnet_train_scratch_floats(M)), device="cuda")
    _lib.check(L.ugrid_rgbnet_train_backward(go.data_ptr(), feat.data_ptr(), h1.data_ptr(), h2.data_ptr(), M, K, 12, w0.data_ptr(), w1.data_ptr(), w2.data_ptr(), 128, gk.data_ptr(), *[t.data_ptr() for t in g], sc.data_ptr(), st), "b")
    gh2 = sc[:M * 128].view(M, 128); gh1 = sc[M * 128:2 * M * 128].view(M, 128)
    rgh2 = (go.double() @ w2.double()) * (rh2 > 0); rgh1 = (rgh2 @ w1.double()) * (rh1 > 0); rgk = (rgh1 @ w0.double())[:, :12]
    e2 = (gh2 - rgh2).abs().amax(1); e1 = (gh1 - rgh1).abs().amax(1); ek = (gk - rgk).abs().amax(1)
    print("  bwd gh2 %.2e gh1 %.2e gk %.2e" % (float(e2.max()), float(e1.max()), float(ek.max())), "bad rows gh1:", torch.nonzero(e1 > 1e-4).flatten()[:10].tolist(), "gk:", torch.nonzero(ek > 1e-4).flatten()[:10].tolist(), int((ek > 1e-4).sum()))
    rw = [rgh1.t() @ feat.double(), rgh1.sum(0), rgh2.t() @ rh1, rgh2.sum(0), go.double().t() @ rh2, go.double().sum(0)]
    print("  wgrad rel", ["%.1e" % float((a - b).abs().max() / b.abs().max()) for a, b in zip(g, rw)])

# timing at the training step's size
M = 83663
feat = torch.randn(M, 39, device="cuda"); go = torch.randn(M, 3, device="cuda")
h1 = torch.empty(M, 128, device="cuda"); h2 = torch.empty(M, 128, device="cuda"); lg = torch.empty(M, 3, device="cuda")
gk = torch.empty(M, 12, device="cuda"); sc = torch.empty(int(L.ugrid_rgbnet_train_scratch_floats(M)), device="cuda")
def fwd():
    _lib.check(L.ugrid_rgbnet_train_forward(feat.data_ptr(), M, 39, w0.data_ptr(), b0.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), 128, h1.data_ptr(), h2.data_ptr(), lg.data_ptr(), st), "f")
def bwd():
    _lib.check(L.ugrid_rgbnet_train_backward(go.data_ptr(), feat.data_ptr(), h1.data_ptr(), h2.data_ptr(), M, 39, 12, w0.data_ptr(), w1.data_ptr(), w2.data_ptr(), 128, gk.data_ptr(), *[t.data_ptr() for t in g], sc.data_ptr(), st), "b")
for name, fn in (("forward", fwd), ("backward", bwd)):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize()
    print("M=%d %s %.1f us" % (M, name, e0.elapsed_time(e1) / 50 * 1e3))
