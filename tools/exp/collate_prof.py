import os, sys, time, torch, cProfile, pstats
sys.path.insert(0, os.getcwd())
import yolat_vectorgraphicsrecognition_amd as yv
item, _, _, _ = yv.config("2")
for i in range(5): yv.collate_to_device([item])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(50): b, sl = yv.collate_to_device([item])
torch.cuda.synchronize()
print("collate_to_device: %.1f us" % ((time.perf_counter() - t0) / 50 * 1e6))
pr = cProfile.Profile(); pr.enable()
for i in range(50): yv.collate_to_device([item])
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
