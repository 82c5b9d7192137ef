import importlib, os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
api = importlib.import_module("21cmfast_amd.grid_api")
S = importlib.import_module("21cmfast_amd.structs")
n = 128; L = 1.5 * n
rng = np.random.default_rng(3)
f = (0.4 * rng.standard_normal((n, n, n))).astype(np.float32)
radii = [0.5, 0.95, 2.0, 6.0, 19.0]
F = np.fft.rfftn(f.astype(np.float64))
dk = 2 * np.pi / L
ki = np.fft.fftfreq(n, 1.0 / n)
kx = (ki * dk).astype(np.float32)[:, None, None]; ky = (ki * dk).astype(np.float32)[None, :, None]
kz = (np.arange(n // 2 + 1) * dk).astype(np.float32)[None, None, :]
ksq = ((kx * kx + ky * ky).astype(np.float32) + kz * kz).astype(np.float32)
k = np.sqrt(ksq.astype(np.float64))
spec = S.rbox_spec(n, L, radii, filter_type=0, min_value=-10.0, const_factor=1.0)
d = torch.from_numpy(f).cuda()
os.environ["C21CM_WINDOWS"] = "table"
tab = api.fill_Rbox_grids(spec, d)["result"].cpu().numpy().astype(np.float64)
del os.environ["C21CM_WINDOWS"]
ev = api.fill_Rbox_grids(spec, d)["result"].cpu().numpy().astype(np.float64)
for i, R in enumerate(radii):
    if R <= spec.cell_radius:
        continue
    x = (k * np.float64(np.float32(R))).astype(np.float32).astype(np.float64)
    xs = np.where(x < 1e-4, 1.0, x)
    Wd = np.where(x < 1e-4, 1 - x * x / 10, 3.0 / xs**3 * (np.sin(xs) - np.cos(xs) * xs))
    truth = np.fft.irfftn(F * Wd, s=(n, n, n))
    truth_f = np.fft.irfftn(F * Wd.astype(np.float32).astype(np.float64), s=(n, n, n))
    rms = truth.std()
    print(f"R={R:5.2f}: (tab-truth)/rms {np.std(tab[i]-truth)/rms:.2e}  (ev-truth)/rms {np.std(ev[i]-truth)/rms:.2e}  "
          f"(floatW-truth)/rms {np.std(truth_f-truth)/rms:.2e}  (ev-tab)/rms {np.std(ev[i]-tab[i])/rms:.2e}")
