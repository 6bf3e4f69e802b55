"""Would a 2-product fp16 rgbnet be accurate enough?     CPU, numpy, no GPU needed.

The shade kernel's fp16x2 arithmetic splits weights and activations into two fp16 parts each (W = Wh + Wl, x = xh + xl, power-of-two
scaled) and sums the three products Wh.xh + Wh.xl + Wl.xh on the matrix cores: fp32 quality (2^-22).  Dropping one cross term
("weights split, activations single fp16", or the converse) would save a third of the MFMAs.  This script evaluates the rgbnet of
the S1 bench (nn.Linear default init, 39-128-128-3) in float64, in the 3-product form and in both 2-product forms, on random
inputs of the bench's statistics, of trained-size features, and with trained-size weights, and prints the worst change of an
output colour.

Result (profiles/r04/two_product_rgbnet_sim.txt): 3 products 2e-8 .. 6e-7; 2 products 2.3e-5 (S1), 6.5e-5 (|k0| ~ 2), 1.2e-3
(weights x 3) -- between a quarter and twelve times the whole 1e-4 budget of the rendered colour, depending on numbers the kernel
cannot bound: the rigorous (interval) bound of the same quantity is ~1e-2 for S1's network, two orders above what happens, so a
guard based on it never enables the mode and a guard based on anything else is not a guard.  The mode is not built."""
import math

import numpy as np


def main():
    rng = np.random.default_rng(0)
    C, pe = 12, 4
    dims = [C + 3 + 6 * pe, 128, 128, 3]
    ws, bs = [], []
    for i in range(3):
        b = 1 / math.sqrt(dims[i])
        ws.append(rng.uniform(-b, b, (dims[i + 1], dims[i])).astype(np.float32))
        bs.append(rng.uniform(-b, b, dims[i + 1]).astype(np.float32))
    bs[2][:] = 0

    def f16(x):
        return x.astype(np.float16).astype(np.float64)

    def split(x):
        h = f16(x)
        return h, f16(x - h)

    def run(feat_std, wscale=1.0, N=200000):
        feat = rng.normal(0, feat_std, (N, C))
        v = rng.normal(size=(N, 3))
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        emb = [v] + [np.sin(v * 2 ** k) for k in range(pe)] + [np.cos(v * 2 ** k) for k in range(pe)]
        x = np.concatenate([feat] + emb, 1).astype(np.float32).astype(np.float64)
        W = [w.astype(np.float64) * wscale for w in ws]
        B = [b.astype(np.float64) for b in bs]

        def net(mode):
            h = x
            for li in range(2):
                Wh, Wl = split(W[li])
                if mode == "exact":
                    z = h @ W[li].T
                elif mode == "3 products":
                    hh, hl = split(h)
                    z = hh @ Wh.T + hl @ Wh.T + hh @ Wl.T
                elif mode == "2 products, activations single":
                    hh = f16(h)
                    z = hh @ Wh.T + hh @ Wl.T
                else:  # "2 products, weights single"
                    hh, hl = split(h)
                    z = hh @ Wh.T + hl @ Wh.T
                h = np.maximum(z + B[li], 0)
            lo = h @ W[2].T + B[2]
            return 1 / (1 + np.exp(-lo)), lo
        ex, le = net("exact")
        for m in ("3 products", "2 products, activations single", "2 products, weights single"):
            r, l = net(m)
            print("|k0| std %-5g weights x %-3g %-34s max |d rgb| %.2e   rms %.2e   max |d logit| %.2e   (|logit| <= %.2f)"
                  % (feat_std, wscale, m, np.abs(r - ex).max(), np.sqrt(((r - ex) ** 2).mean()), np.abs(l - le).max(), np.abs(le).max()))
    run(0.25)
    run(2.0)
    run(2.0, 3.0, 50000)


if __name__ == "__main__":
    main()
