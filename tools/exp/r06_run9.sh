timeout 2400 python -m pytest tests/test_gpu_model.py tests/test_gpu_dist.py tests/test_gpu_loader.py -x -q 2>&1 | tail -8
python - <<'PY'
import sys, time, torch
sys.path.insert(0, "tests")
import golden_util as gu
import yolat_vectorgraphicsrecognition_amd as yv
from yolat_vectorgraphicsrecognition_amd import architecture as A
data, slices = yv.synth_batch(1, 11, num_proposals=2000, nodes_lo=4, nodes_hi=40, edge_factor=1.2, with_roots=True)
model = gu.fill_state_(yv.SparseCADGCN(yv.Opt(n_classes=17, n_blocks=2, n_blocks_out=2)), 0).cuda().eval()
for flag in (True, False, True, False):
    A.PREDICT_ONE_SUBMISSION = flag
    with torch.no_grad():
        for _ in range(5):
            model.predict(data, slices)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            out = model.predict(data, slices)
        torch.cuda.synchronize()
    print("predict %s: %.3f ms per call, %d rows" % ("one submission" if flag else "two passes", (time.perf_counter() - t0) / 50 * 1e3, out[0].shape[0]))
PY
