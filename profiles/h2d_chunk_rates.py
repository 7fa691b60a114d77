"""H2D rate from page-locked host memory by copy size and by concurrency (one or two streams): why the f32 host-buffer
path of bench.py (1.2 MB depth planes) reached 21 GB/s and the u16 path (0.6 MB planes) 41 GB/s in round 2."""
import time
import torch
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
total = 128 << 20
src = torch.empty(total, dtype=torch.uint8).pin_memory()
dst = torch.empty(total, dtype=torch.uint8, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for chunk in (128 << 10, 300 << 10, 600 << 10, 900 << 10, 1200 << 10, 2400 << 10, 8 << 20):
    for nstream in (1, 2):
        n = total // chunk
        best = 0.0
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(n):
                st = s1 if (nstream == 1 or i % 2 == 0) else s2
                with torch.cuda.stream(st):
                    dst[i * chunk:(i + 1) * chunk].copy_(src[i * chunk:(i + 1) * chunk], non_blocking=True)
            torch.cuda.synchronize()
            best = max(best, n * chunk / (time.perf_counter() - t0) / 1e9)
        print("chunk %7d KiB, %d stream(s): %5.1f GB/s" % (chunk >> 10, nstream, best))
