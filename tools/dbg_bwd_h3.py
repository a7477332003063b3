"""GPU box: where do repeated launches of the separate backward kernel differ?  Runs the same step `reps` times from the same
state with the library named by ULTR_HIP_LIB (tools/h3_repro.sh builds the repro variants), snapshots dz_0 / dz_1 (the
backward kernel's outputs in the workspace) and prints, for every element that is not the same in all launches: counts by
row-in-tile (n % 16), by 16-column tile, by workgroup, and the ratio bad / majority value.
    ULTR_HIP_LIB=ultra_pytorch_amd/lib/variants/libultr_h3B.so python tools/dbg_bwd_h3.py [reps]"""
import os, sys, collections, numpy as np, torch
sys.path.insert(0, os.getcwd())
os.environ.setdefault("ULTR_NO_FUSED_FB", "1")
from ultra_pytorch_amd import engine, hip_ops, synthetic
from ultra_pytorch_amd.ranking_model import init_flat_params


def dz_offsets(F, hidden, N):
    K = [F] + list(hidden)
    M = list(hidden) + [1]
    nl = len(K)
    P = sum(2 * K[j] + K[j] * M[j] + M[j] for j in range(nl))
    r4 = lambda x: (x + 3) & ~3
    off = r4((P + 4096 + 63) // 64)
    vlen = sum(2 * k for k in K) + K[-1] + 1
    nrb = (N + 15) // 16
    off = r4(off + max((N + 8) // 9 + 1, nrb) * vlen)
    off = r4(off + vlen)
    out = []
    for j in range(nl - 1):
        out.append(off)
        off = r4(off + N * M[j])
    return out


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    F, hidden, B, L = 136, [256, 256], 256, 10
    if os.environ.get("DBG_CFG3") == "1":
        F, hidden, B, L = 136, [512, 256, 128], 512, 20
    N = B * L
    shape = hip_ops.DnnShape(F, hidden, "elu")
    rng = np.random.RandomState(5)
    feats, ids, y = synthetic.make_batch(rng, B, L, F)
    ipw = np.asarray(synthetic.load_ipw(), np.float32)
    p0 = init_flat_params(shape, seed=3).numpy()
    dev = lambda a, dt=torch.float32: torch.as_tensor(a).to("cuda", dt)
    eng = engine.StepEngine(shape, B, L, torch.device("cuda"), algo="softmax", learning_rate=0.05, max_gradient_norm=5.0)
    f, i, yy, tab = dev(feats), dev(ids, torch.int32), dev(y), dev(ipw)
    offs = dz_offsets(F, hidden, N)
    snaps = [[] for _ in offs]
    dbg = None
    for r in range(reps):
        params, state = dev(p0.copy()), dev(np.zeros_like(p0))
        eng.train_step(params, state, f, feats.shape[0], i, yy, ipw_table=tab)
        torch.cuda.synchronize()
        if hasattr(shape.lib, "ultr_dbg_read"):
            import ctypes
            buf = np.empty(1 << 21, np.float32)
            shape.lib.ultr_dbg_read(buf.ctypes.data_as(ctypes.c_void_p))
            dbg = (dbg or []) + [buf]
        for j, o in enumerate(offs):
            snaps[j].append(eng.bwd_ws[o:o + N * hidden[j]].view(N, hidden[j]).cpu().numpy().copy())
    if dbg is not None:  # H3_DBG_DUMP build: du_1 as the row pass read it, and the scales every wave's epilogue read
        d = np.stack(dbg)
        snaps.append([x[:N * hidden[0]].reshape(N, hidden[0]) for x in d])
        nwg = (N + 15) // 16
        osd = d[:, 700000:700000 + nwg * 128].reshape(len(d), nwg, 8, 16)
        same_w = (osd == osd[:, :, :1, :]).all(axis=2)  # do the 8 waves of a workgroup agree on the 16 scales?
        print("scales read by the epilogue: waves of a workgroup disagree in %d (launch, workgroup, row) cases; rows: %s" %
              (int((~same_w).sum()), sorted(collections.Counter(np.nonzero(~same_w)[2].tolist()).items())))
        med_os = np.median(osd, axis=0)
        bad_os = osd != med_os[None]
        print("scales off their majority value over launches: %d; rows %s; waves %s" %
              (int(bad_os.sum()), sorted(collections.Counter(np.nonzero(bad_os)[3].tolist()).items()),
               sorted(collections.Counter(np.nonzero(bad_os)[2].tolist()).items())))
        if bad_os.any():
            l_, w_, v_, r_ = [x[:5] for x in np.nonzero(bad_os)]
            for k in range(len(l_)):
                print("   e.g. launch %d wg %d wave %d row %d: read %.6g, majority %.6g" % (l_[k], w_[k], v_[k], r_[k], osd[l_[k], w_[k], v_[k], r_[k]], med_os[w_[k], v_[k], r_[k]]))
        # the epilogue's register results, rearranged as du[row, col]: wave w = chunk w, lane (q, i), tile t, register r -> row 4q + r, col 32w + 2i + t
        epi = d[:, (1 << 20):(1 << 20) + nwg * 8 * 64 * 8].reshape(len(d), nwg, 8, 4, 16, 2, 4)  # [launch, wg, wave, q, i, t, r]
        epi = epi.transpose(0, 1, 3, 6, 2, 4, 5).reshape(len(d), nwg * 16, 256)[:, :N]  # [launch, wg * 16 + 4q + r, 32w + 2i + t]
        du_lds = np.stack(snaps[-1])
        if not epi.any():
            epi = du_lds  # H3_DBG_DUMP=1 build: no register dump
        neq = epi != du_lds
        print("epilogue registers vs du as the row pass read it from LDS: %d elements differ; rows (n %% 16) %s; column parity (tile t) %s" %
              (int(neq.sum()), sorted(collections.Counter((np.nonzero(neq)[1] % 16).tolist()).items()),
               sorted(collections.Counter((np.nonzero(neq)[2] % 2).tolist()).items())))
        med_e = np.median(epi, axis=0)
        bad_e = epi != med_e[None]
        print("epilogue registers off their majority value over launches: %d; rows %s; tile t %s" %
              (int(bad_e.sum()), sorted(collections.Counter((np.nonzero(bad_e)[1] % 16).tolist()).items()),
               sorted(collections.Counter((np.nonzero(bad_e)[2] % 2).tolist()).items())))
        l_, n_, c_ = [x[:8] for x in np.nonzero(bad_e)]
        for k in range(len(l_)):
            print("   e.g. launch %d row %d col %d: register %.8g, LDS %.8g, majority %.8g (ratio reg/majority %.6g)" %
                  (l_[k], n_[k], c_[k], epi[l_[k], n_[k], c_[k]], du_lds[l_[k], n_[k], c_[k]], med_e[n_[k], c_[k]], epi[l_[k], n_[k], c_[k]] / med_e[n_[k], c_[k]]))
        offs = offs + [None]
        names = ["dz_%d" % j for j in range(len(offs) - 1)] + ["du_1 as read by the row pass"]
    else:
        names = ["dz_%d" % j for j in range(len(offs))]
    hidden = hidden + [hidden[0]]
    for j in range(len(offs)):
        a = np.stack(snaps[j])  # [reps, N, M]
        med = np.median(a, axis=0)
        bad = a != med[None]
        print("%s: %d of %d launches have an element off the majority value; %d distinct bad (launch, element) pairs" %
              (names[j], int(bad.any(axis=(1, 2)).sum()), reps, int(bad.sum())))
        if not bad.any():
            continue
        rr, nn, cc = np.nonzero(bad)
        if names[j].startswith("du_1") and dbg is not None:
            # which scale would explain the bad value?  bad / majority against the ratios of the neighbouring rows' scales
            osr = osd[0, :, 0, :]  # [wg, 16] (stable over launches, identical in all waves)
            for k in range(min(12, len(rr))):
                wg, r16 = nn[k] // 16, nn[k] % 16
                print("   launch %d row %d col %d: bad %.8g, majority %.8g, ratio %.6g; scales of rows %d..%d: %s" %
                      (rr[k], nn[k], cc[k], a[rr[k], nn[k], cc[k]], med[nn[k], cc[k]], a[rr[k], nn[k], cc[k]] / med[nn[k], cc[k]],
                       r16 & ~3, (r16 & ~3) + 3, osr[wg, (r16 & ~3):(r16 & ~3) + 4]))
        print("   by row in tile (n % 16):", sorted(collections.Counter((nn % 16).tolist()).items()))
        print("   by 16-column tile      :", sorted(collections.Counter((cc // 16).tolist()).items()))
        print("   by 32-column chunk     :", sorted(collections.Counter((cc // 32).tolist()).items()))
        print("   workgroups touched     : %d of %d" % (len(set((nn // 16).tolist())), (N + 15) // 16))
        print("   bad elements per (launch, row):", sorted(collections.Counter(collections.Counter(zip(rr.tolist(), nn.tolist())).values()).items()))
        ratio = a[bad] / med[nn, cc]
        lr = np.log2(np.abs(ratio[np.isfinite(ratio) & (ratio != 0)]))
        if lr.size == 0:
            print('   every bad value is 0, inf or nan:', a[bad][:8])
            continue
        print("   log2 |bad / majority|  : min %.4f  median %.4f  max %.4f;  within 1e-3 of an integer: %.1f %%" %
              (lr.min(), np.median(lr), lr.max(), 100.0 * np.mean(np.abs(lr - np.round(lr)) < 1e-3)))
        rel = np.abs(a[bad] - med[nn, cc]) / (np.abs(med[nn]).max(axis=1) + 1e-30)
        print("   |bad - majority| / max|row|: median %.2e  max %.2e" % (np.median(rel), rel.max()))
        for k in range(min(6, len(rr))):
            print("   e.g. launch %d row %d (n %% 16 = %d) col %d: %.8g vs majority %.8g" % (rr[k], nn[k], nn[k] % 16, cc[k], a[rr[k], nn[k], cc[k]], med[nn[k], cc[k]]))


if __name__ == "__main__":
    main()
