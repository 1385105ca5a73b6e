"""Measurement build only (FASTKV_BUILD_DIR=build_x_stamp FASTKV_CXXFLAGS=-DFK_STAMP python fastkv_amd/_build.py): WHO arrives last at the
hand-offs of the rolling launch of score_fused (VERDICT r05 next #3a).  Eight 32k layers in one rolling launch; the stamp table keeps the
last four entries (1024 waves each).  Per entry and unit (KV head: 32 workgroups): when each workgroup published its row maxima (slot 22),
its row sums (slot 4), how long its phase A took, where it ran (XCD, compute unit) and which span of the row it owns -- and, over all
units, whether the LAST arrivals are the same few: the span index (the last span holds the window tile and the ragged end), the XCD, the
start time (a late dispatch) or the phase-A time (contention) of the stragglers.  Usage: stamp_arrivals.py [entries=8]"""
import ctypes, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from fastkv_amd import ops, _lib
dev = torch.device('cuda:0')
H, Hkv, D, W, S, B = 32, 8, 128, 8, 32768, int(sys.argv[1]) if len(sys.argv) > 1 else 8
lib = _lib.load()
sets = [(torch.randn(B, S, H, D, device=dev, dtype=torch.float16).transpose(1, 2), torch.randn(B, S, Hkv, D, device=dev, dtype=torch.float16).transpose(1, 2)) for _ in range(3)]
for i in range(7):
    ops.scores(*sets[i % 3], W, 7, 'maxpool', want_tsp=False)
torch.cuda.synchronize()
buf = np.zeros(4096 * 48, dtype=np.uint64)
lib.fastkv_debug_read_fused_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.size))
st = buf.reshape(4096, 48)
t = st.astype(np.int64)
ent = (st[:, 41] >> np.uint64(32)).astype(np.int64); unit = ((st[:, 41] >> np.uint64(16)) & np.uint64(0xffff)).astype(np.int64) % Hkv; span = (st[:, 41] & np.uint64(0xffff)).astype(np.int64)
xcc = ((st[:, 40] >> np.uint64(32)) & np.uint64(7)).astype(np.int64); hw = (st[:, 40] & np.uint64(0xffffffff)).astype(np.int64); cu = (hw >> 8) & 0xff
ok = t[:, 14] > 0
print(f"contraction {os.environ.get('FASTKV_CONTRACTION', 'fmaf (default)')}; rolling launch of {B} entries of {S} tokens; entries in the table: {sorted(set(ent[ok].tolist()))}")
last_span = collections.Counter(); last_xcc = collections.Counter(); rows = []
behind = []                                                      # every workgroup: its arrival at hand-off 1 behind its head's FIRST arrival (us)
for e in sorted(set(ent[ok].tolist())):
    for u in range(Hkv):
        m = ok & (ent == e) & (unit == u)
        if m.sum() == 0:
            continue
        # per workgroup: the latest of its four waves
        wg = {}
        for i in np.nonzero(m)[0]:
            k_ = int(span[i])
            a = wg.setdefault(k_, dict(start=t[i, 0], a_end=0, pub_max=0, max_known=0, pub_sum=0, sum_known=0, end=0, xcc=int(xcc[i]), cu=int(cu[i])))
            a["start"] = min(a["start"], t[i, 0]); a["a_end"] = max(a["a_end"], t[i, 21]); a["pub_max"] = max(a["pub_max"], t[i, 22])
            a["max_known"] = max(a["max_known"], t[i, 3]); a["pub_sum"] = max(a["pub_sum"], t[i, 4]); a["sum_known"] = max(a["sum_known"], t[i, 7]); a["end"] = max(a["end"], t[i, 14])
        spans = sorted(wg)
        pm = np.array([wg[s_]["pub_max"] for s_ in spans]) / 100.0; ps = np.array([wg[s_]["pub_sum"] for s_ in spans]) / 100.0
        stt = np.array([wg[s_]["start"] for s_ in spans]) / 100.0; pa = pm - stt
        mk = np.array([wg[s_]["max_known"] for s_ in spans]) / 100.0
        behind.extend((pm - pm.min()).tolist())
        i_last = int(np.argmax(pm))
        last_span[spans[i_last]] += 1; last_xcc[wg[spans[i_last]]["xcc"]] += 1
        rows.append(dict(e=e, u=u, n=len(spans), spread_max=float(pm.max() - np.median(pm)), spread_sum=float(ps.max() - np.median(ps)), last_span=spans[i_last],
                         last_started_late=float(stt[i_last] - np.median(stt)), last_phaseA=float(pa[i_last]), med_phaseA=float(np.median(pa)),
                         wait_med=float(np.median(mk - pm)), wait_max=float((mk - pm).max()), resident_med=float(np.median(np.array([wg[s_]["end"] for s_ in spans]) / 100.0 - stt))))
print("per (entry, unit): last arrival at hand-off 1 behind the unit's MEDIAN arrival (us), the same at hand-off 2, which span it was, how much later than the median it STARTED, its phase-A time against the unit's median, median / longest wait at hand-off 1, median residency")
for r in rows:
    print(f"  e{r['e']} u{r['u']} ({r['n']} wgs): +{r['spread_max']:5.2f} / +{r['spread_sum']:5.2f} us   last = span {r['last_span']:2d}  started {r['last_started_late']:+5.2f}  phase A {r['last_phaseA']:5.2f} vs {r['med_phaseA']:5.2f}   wait {r['wait_med']:5.2f} / {r['wait_max']:5.2f}   resident {r['resident_med']:5.1f}")
sm = np.array([r["spread_max"] for r in rows]); lt = np.array([r["last_started_late"] for r in rows]); pa = np.array([r["last_phaseA"] - r["med_phaseA"] for r in rows])
print(f"over {len(rows)} (entry, unit) pairs: last arrival behind the median {np.median(sm):.2f} us (max {sm.max():.2f}); of that, late START {np.median(lt):+.2f} us, longer PHASE A {np.median(pa):+.2f} us")
hb = np.histogram(np.array(behind), bins=[0, 1, 2, 3, 4, 5, 6, 8, 10, 15, 1000])[0]
print("histogram over all", len(behind), "workgroups -- arrival at hand-off 1 behind the head's FIRST arrival, bins [0,1) [1,2) [2,3) [3,4) [4,5) [5,6) [6,8) [8,10) [10,15) [15,inf) us:", hb.tolist())
print("span of the last arrival:", dict(sorted(last_span.items())), " XCD of the last arrival:", dict(sorted(last_xcc.items())))
