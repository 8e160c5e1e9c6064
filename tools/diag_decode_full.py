import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import test_decode_full as T
z = np.load(T.GOLD)
got, enc, d = T.decode(torch.device("cuda:0"), "fp32")
print("lens equal", np.array_equal(got["lens"], z["lens"]))
for b in range(4):
    for j in range(16):
        L = int(z["lens"][b, j])
        same = got["lens"][b, j] == L and np.array_equal(got["hyps"][b, j, :L], z["hyps"][b, j, :L])
        if not same:
            # is it a permutation within the list?
            where = [k for k in range(16) if got["lens"][b, k] == L and np.array_equal(got["hyps"][b, k, :L], z["hyps"][b, j, :L])]
            dpos = [int(i) for i in np.nonzero(got["hyps"][b, j, :L] != z["hyps"][b, j, :L])[0][:5]] if got["lens"][b, j] == L else None
            print("b%d j%d differs: ref score %.6f got score %.6f; ref hyp found at our rank %s; first diff positions %s" % (
                b, j, z["scores"][b, j], got["scores"][b, j], where, dpos))
print("max score diff", np.abs(got["scores"] - z["scores"]).max())
d.fused_search = False
got2, _, _ = T.decode(torch.device("cuda:0"), "fp32")
