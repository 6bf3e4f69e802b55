import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unboundednerfpytorch_amd import _lib
L = _lib.load()
M = 83663
torch.manual_seed(0)
feat = torch.randn(M, 39, device="cuda")
w0 = torch.randn(128, 39, device="cuda") * 0.1; b0 = torch.randn(128, device="cuda") * 0.1
w1 = torch.randn(128, 128, device="cuda") * 0.1; b1 = torch.randn(128, device="cuda") * 0.1
w2 = torch.randn(3, 128, device="cuda") * 0.1; b2 = torch.randn(3, device="cuda") * 0.1
h1 = torch.empty(M, 128, device="cuda"); h2 = torch.empty(M, 128, device="cuda"); lg = torch.empty(M, 3, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def fwd():
    _lib.check(L.ugrid_rgbnet_train_forward(feat.data_ptr(), M, 39, w0.data_ptr(), b0.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), 128, h1.data_ptr(), h2.data_ptr(), lg.data_ptr(), st), "f")
for _ in range(5): fwd()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): fwd()
e1.record(); torch.cuda.synchronize()
print(os.environ.get("UGRID_LIB", "product"), "forward %.1f us" % (e0.elapsed_time(e1) / 50 * 1e3))
