#!/usr/bin/env python3
"""Host ceiling of the configs[4] chain per NODE: P processes -- one per GPU rank -- each writing OBJ-sized
files to one tmpfs the way the library's write-behind threads do (mesh_writer.hip write_file_parallel: T threads of pwrite
over one file, from a buffer in host memory), no GPU involved.  Prints files/s and GB/s per process count.

    python3 tools/node_write_ceiling.py [dir=/dev/shm] [MB per file=110] [seconds per point=4]
"""
import multiprocessing as mp
import os
import sys
import threading
import time


def writer(rank, directory, nbytes, threads, seconds, out):
    buf = bytes(bytearray(os.urandom(1 << 20)) * (nbytes >> 20))
    chunk = (len(buf) + threads - 1) // threads
    done = 0
    t_end = time.time() + seconds
    t0 = time.time()
    while time.time() < t_end:
        path = os.path.join(directory, f"me_ceiling_{rank}_{done & 1}.obj")
        fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
        try:
            os.ftruncate(fd, len(buf))

            def part(i):
                lo = i * chunk
                view = memoryview(buf)[lo:lo + chunk]
                off = 0
                while off < len(view):
                    off += os.pwrite(fd, view[off:off + (8 << 20)], lo + off)
            ts = [threading.Thread(target=part, args=(i,)) for i in range(threads)]
            for t in ts:
                t.start()
            for t in ts:
                t.join()
        finally:
            os.close(fd)
        done += 1
    dt = time.time() - t0
    for k in (0, 1):
        try:
            os.unlink(os.path.join(directory, f"me_ceiling_{rank}_{k}.obj"))
        except OSError:
            pass
    out.put((rank, done, dt))


def main():
    directory = sys.argv[1] if len(sys.argv) > 1 else "/dev/shm"
    mb = int(sys.argv[2]) if len(sys.argv) > 2 else 110
    seconds = float(sys.argv[3]) if len(sys.argv) > 3 else 4.0
    nbytes = mb << 20
    print(f"# {mb} MB files on {directory}, {os.cpu_count()} host cores visible, {seconds:.0f} s per point")
    for threads in (8, 2):
        for procs in (1, 2, 4, 8):
            q = mp.Queue()
            ps = [mp.Process(target=writer, args=(r, directory, nbytes, threads, seconds, q)) for r in range(procs)]
            for p in ps:
                p.start()
            res = [q.get() for _ in ps]
            for p in ps:
                p.join()
            files_s = sum(d / dt for _, d, dt in res)
            print(f"{procs} process(es) x {threads} pwrite threads: {files_s:6.1f} files/s in total "
                  f"({files_s / procs:5.1f} per process), {files_s * nbytes / 1e9:5.1f} GB/s", flush=True)


if __name__ == "__main__":
    main()
