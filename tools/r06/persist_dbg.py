import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tools", "r06"))
import s16_probe as sp, persist_ab as ab
from ddnm_amd import ops
s = [x for x in ab.SHAPES if x[0] == "c128_128_256_b3"][0]
t = sp.make(*s); t["badd"] = torch.randn(s[1], s[4], device="cuda"); t["amax"] = None
a = ab.call(t, True)(); b = ab.call(t, False)()
torch.cuda.synchronize()
B, Cout = s[1], s[4]
sa = a.stats.view(B * a.tiles, Cout, 2); sb = b.stats.view(B * b.tiles, Cout, 2)
d = (sa - sb).abs()
print("tiles", a.tiles, "max abs", d.max().item(), "rel", (d.max() / sa.abs().max()).item())
bad = (d > 0).any(2).any(1)
print("tiles differing:", bad.sum().item(), "of", bad.numel(), "first few idx", bad.nonzero()[:10].flatten().tolist())
i = bad.nonzero()[0].item()
print("tile", i, "one-tile", sa[i, :4].tolist(), "persist", sb[i, :4].tolist())
ch = (d[i] > 0).any(1).nonzero().flatten().tolist()
print("channels differing in that tile:", len(ch), ch[:16])
# the true sums
o = a.t.view(B, 256 // 8, 8, 256 // 32, 32, Cout)
m = i; img = m // 256; tt = m % 256; ty, tx = tt // 8, tt % 8
tile = o[img, ty, :, tx]
print("true sum ch0..1", tile[..., 0].double().sum().item(), (tile[..., 0].double() ** 2).sum().item(), tile[..., 1].double().sum().item())
