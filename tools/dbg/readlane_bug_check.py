"""Does round 3's small_n_bwd_weight_kernel (v_readlane inside the divergent `if (live)`) give wrong sums at the headline's shapes?
S = db^T a2 for one 864-column chunk, N = 32768 rows, A + 1 = 4 heads: old library vs current library vs fp64."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-a2c_amd")]
import torch
from a2c_amd import ops
old = ctypes.CDLL(os.path.join(ROOT, "tools", "dbg", "libold_gemm_r3.so"))
P, I64 = ctypes.c_void_p, ctypes.c_int64
old.a2c_gemm_f32.argtypes = [ctypes.c_int, ctypes.c_int, I64, I64, I64, P, I64, P, I64, P, I64, P, ctypes.c_int, P, I64, ctypes.c_int, ctypes.c_int, P, ctypes.c_size_t, P]
old.a2c_gemm_ws_bytes.restype = ctypes.c_size_t
old.a2c_gemm_ws_bytes.argtypes = [I64, I64, ctypes.c_int]
torch.manual_seed(0)
N, F, Kc, A1 = 32768, 2592, 864, 4
db = torch.randn(N, A1, device="cuda")
a2 = torch.relu(torch.randn(N, F, device="cuda"))
want = (db.double().t() @ a2[:, :Kc].double())
ws = torch.zeros(old.a2c_gemm_ws_bytes(A1, Kc, 1) // 4 + 16, device="cuda")
out_old = torch.zeros(A1, Kc, device="cuda")
rc = old.a2c_gemm_f32(1, 0, A1, Kc, N, db.data_ptr(), A1, a2.data_ptr(), F, out_old.data_ptr(), Kc, None, 0, None, 0, 0, 1, ws.data_ptr(), ws.numel() * 4, None)
torch.cuda.synchronize()
out_new = torch.zeros(A1, Kc, device="cuda")
ops.gemm(1, 0, A1, Kc, N, db.data_ptr(), A1, a2.data_ptr(), F, out_new.data_ptr(), Kc, ws=ws)
torch.cuda.synchronize()
sc = float(want.abs().max())
for name, o in (("round-3 kernel", out_old), ("current kernel", out_new)):
    e = (o.double() - want).abs() / sc
    print(f"{name}: rc={rc} max rel err columns 0..767: {float(e[:, :768].max()):.2e}   columns 768..863 (last, partly live wave): {float(e[:, 768:].max()):.2e}")
