import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, reflexiv_amd
rfx = reflexiv_amd.Reflexiv(0)
k, G, n_reads, L = 63, 30_000, 4000, 150
seed = 31 + k
wpr = (L + 31) // 32
dg = torch.empty((G + 31) // 32, dtype=torch.int64, device="cuda"); dw = torch.empty(n_reads * wpr, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
rfx.synth_genome_dev(seed, G, dg.data_ptr()); rfx.synth_reads_dev(seed, dg.data_ptr(), G, 0, n_reads, L, wpr, dw.data_ptr()); rfx.sync()
owners = int(os.environ.get("OWNERS", "4"))
doff = torch.empty(owners + 1, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
def records():
    need, h = rfx.bucket_wide_records_by_owner_dev(dw.data_ptr(), n_reads, wpr, L, k, owners, 0, 0, doff.data_ptr())
    out = torch.full((3 * need,), -1, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    nrec, h = rfx.bucket_wide_records_by_owner_dev(dw.data_ptr(), n_reads, wpr, L, k, owners, out.data_ptr(), need, doff.data_ptr())
    r = out.cpu().numpy().view(np.uint64).reshape(nrec, 3)
    return r[np.lexsort((r[:, 2], r[:, 1], r[:, 0]))], h
os.environ["RFX_SK_DESC"] = "0"
ref, href = records()
for mode in ("0", "1", "1", "1"):
    os.environ["RFX_SK_DESC"] = mode
    got, h = records()
    same = got.shape == ref.shape and np.array_equal(got, ref)
    print("desc", mode, "records", got.shape, "identical multiset:", same, "owner offsets equal:", np.array_equal(h, href))
    if not same:
        rs = set(map(tuple, ref.tolist())); gs = set(map(tuple, got.tolist()))
        extra = sorted(gs - rs); missing = sorted(rs - gs)
        print("  rows only in got:", len(extra), " only in ref:", len(missing))
        byw = {}
        for row in missing:
            for i in range(3): byw.setdefault((i, row[i]), []).append(row)
        shown = 0
        for row in extra:
            cands = [c for i in range(3) for c in byw.get((i, row[i]), [])]
            if cands and shown < 12:
                shown += 1
                print("   got", [hex(x) for x in row], "\n   ref", [hex(x) for x in cands[0]])
    if False:
        bad = np.nonzero((got != ref).any(axis=1))[0]
        print("  differing rows:", len(bad), "first:", [hex(int(x)) for x in got[bad[0]]], "vs", [hex(int(x)) for x in ref[bad[0]]])
        unw = (got == np.uint64(0xFFFFFFFFFFFFFFFF)).all(axis=1).sum()
        print("  rows never written:", int(unw))
