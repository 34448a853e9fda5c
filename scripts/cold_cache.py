#!/usr/bin/env python
"""Evicts the shared libraries of torch / ROCm / this repo from the page cache (POSIX_FADV_DONTNEED;
/proc/sys/vm/drop_caches when writable), to reproduce the state of a FRESH GPU box, where the first
use of every kernel pages its code object in from disk (FID-10k sampling took 30 s on
the driver's fresh lease and 0.3 s on a warm box)."""
import glob
import os
import sys


def main():
    try:
        os.sync()
        with open("/proc/sys/vm/drop_caches", "w") as f:
            f.write("3\n")
        print("drop_caches: ok")
    except OSError as e:
        print("drop_caches: %s" % e)
    import importlib.util
    spec = importlib.util.find_spec("torch")
    roots = [os.path.join(os.path.dirname(spec.origin), "lib"), "/opt/rocm/lib",
             os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "compare_gan_amd", "lib")]
    n = tot = 0
    for root in roots:
        for path in glob.glob(os.path.join(root, "**", "*"), recursive=True):
            if not os.path.isfile(path) or os.path.islink(path):
                continue
            try:
                fd = os.open(path, os.O_RDONLY)
                os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
                tot += os.fstat(fd).st_size
                os.close(fd)
                n += 1
            except OSError:
                pass
    print("fadvise(DONTNEED) on %d files, %.1f GB" % (n, tot / 1e9))


if __name__ == "__main__":
    sys.exit(main())
