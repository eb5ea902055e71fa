"""Lane use of a home-walk backward: a thread group per home pixel walks that home's samples.
Simulates, for one synthetic image, the number of walk steps (max count over the homes of a wave,
per chunk) against the samples processed."""
import sys, math
import numpy as np, torch
sys.path.insert(0, '/root/repo')
from instaorder_amd import synthetic

def run(image=1024, dist='model', TH=16, TW=16, chunk=2048, homes_per_wave=16, wave_shape='row', heads=8):
    shapes = synthetic.encoder_shapes(image)
    inp = synthetic.make_inputs(1, shapes, dist=dist, seed=0)
    loc = inp['sampling_locations'][0].numpy()   # [Lq, M, L, P, 2]
    Lq, M, L, P, _ = loc.shape
    tot_samples = 0; tot_steps = 0; items = 0
    per_level = []
    for l, (H, W) in enumerate(shapes):
        x = np.float32(loc[:, :, l, :, 0]) * np.float32(W) - np.float32(0.5)
        y = np.float32(loc[:, :, l, :, 1]) * np.float32(H) - np.float32(0.5)
        ok = (y > -1) & (x > -1) & (y < H) & (x < W)
        hy = np.floor(y).astype(np.int64) + 1
        hx = np.floor(x).astype(np.int64) + 1
        nty = (H + 1 + TH - 1) // TH; ntx = (W + 1 + TW - 1) // TW
        lvl_samples = 0; lvl_steps = 0
        for m in range(heads):
            o = ok[:, m, :].reshape(-1)
            yy = hy[:, m, :].reshape(-1)[o]; xx = hx[:, m, :].reshape(-1)[o]
            tile = (yy // TH) * ntx + (xx // TW)
            order = np.argsort(tile, kind='stable')      # query order kept inside a tile
            tile = tile[order]; yy = yy[order]; xx = xx[order]
            bounds = np.searchsorted(tile, np.arange(nty * ntx + 1))
            for t in range(nty * ntx):
                a, b = bounds[t], bounds[t + 1]
                items += 1
                if a == b: continue
                ly = yy[a:b] % TH; lx = xx[a:b] % TW
                if wave_shape == 'row':
                    wave = (ly * TW + lx) // homes_per_wave
                    nw = TH * TW // homes_per_wave
                else:   # 4x4 patches
                    wave = (ly // 4) * (TW // 4) + lx // 4
                    nw = (TH // 4) * (TW // 4)
                home = ly * TW + lx
                for c0 in range(0, b - a, chunk):
                    h = home[c0:c0 + chunk]; wv = wave[c0:c0 + chunk]
                    cnt = np.bincount(h, minlength=TH * TW)
                    # max count per wave
                    wmax = np.zeros(nw, dtype=np.int64)
                    hw = (np.arange(TH * TW) // homes_per_wave) if wave_shape == 'row' else \
                         ((np.arange(TH * TW) // TW // 4) * (TW // 4) + (np.arange(TH * TW) % TW) // 4)
                    np.maximum.at(wmax, hw, cnt)
                    lvl_steps += wmax.sum(); lvl_samples += len(h)
        per_level.append((H, W, lvl_samples, lvl_steps, lvl_samples / max(1, lvl_steps * homes_per_wave)))
        tot_samples += lvl_samples; tot_steps += lvl_steps
    return tot_samples, tot_steps, tot_samples / (tot_steps * homes_per_wave), items, per_level

if __name__ == '__main__':
    for dist in ('model', 'uniform'):
        for TH in (8, 16):
            for chunk in (512, 1024, 2048, 4096):
                for ws in ('row', 'patch'):
                    s, st, u, items, pl = run(dist=dist, TH=TH, chunk=chunk, wave_shape=ws)
                    print(dist, 'TH', TH, 'chunk', chunk, ws, 'samples', s, 'steps', st, 'u %.3f' % u, 'items', items,
                          ' '.join('L%dx%d:%.2f' % (h, w, uu) for h, w, _, _, uu in pl), flush=True)
