#!/usr/bin/env python3
"""Benchmark of the MI355X-native JPEG XL VarDCT decode path (BASELINE.json metric: megapixels/sec decode).

A "step" decodes one batch of synthetic frames per GPU: `--batch` copies of a 3840x2160 RGB8 frame encoded at
distance 1.0 (Gaborish + EPF1, one pass), i.e. BASELINE.json configs[1].  The compressed AC sections and all per-frame
tables are resident in HBM before the timed region; each frame has its own device buffers and HIP stream.  A step runs
every stage once over `--batch` frames and leaves RGB8 in HBM: ONE lane-parallel entropy launch over all AC sections of
a frame set, and dequant/IDCT + fused Gaborish/EPF/colour per frame.  The stages are software-pipelined over two frame
sets (while set A is in the latency-bound entropy kernel, set B's coefficients run through the bandwidth-bound
stages), so each step completes `--batch` frames; `--no-pipeline` runs one set strictly in sequence.

Multi-GPU (`--gpus N`, launched by torch.distributed.run, one rank per GPU): frames are independent, so every rank
decodes its own batch (weak scaling) with no data-path collective; RCCL is only used for the barrier and the
max-over-ranks time.  MP/s follows the reference's definition xsize*ysize*1e-6/elapsed (tools/speed_stats.cc:107).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # the few streams in use (two frame sets x two, the copy stream, the shared pool) each on a hardware queue of their own

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured achievable rate


def make_stream(xsize, ysize, distance, seed=177, max_clusters=0, ac_code_mode=0):
    import libjxl_amd as J
    cache = "/tmp/libjxl_amd_bench_%dx%d_d%.2f_s%d_c%d_m%d.jxl" % (xsize, ysize, distance, seed, max_clusters, ac_code_mode)
    if os.path.exists(cache):
        return open(cache, "rb").read()
    img = J.synth_image(xsize, ysize, seed)
    data = J.encode_rgb8(img, distance=distance, strategy_mode=1, max_clusters=max_clusters, ac_code_mode=ac_code_mode)
    try:
        tmp = cache + ".%d" % os.getpid()
        open(tmp, "wb").write(data)
        os.replace(tmp, cache)
    except OSError:
        pass
    return data


def cpu_baseline(data, xsize, ysize, budget_s=20.0):
    """Times the oracle (CPU restatement, kind 'port': scalar code, OpenMP threads over groups and rows) on the same
    stream, on the host cores this process may use."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import jxlo
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    # A timing-only object of the same source, built HERE for this host's CPU: -O3 -march=native as SURVEY.md 8d / BASELINE.md
    # B2 ask (the parity build oracle/libjxlo.so is -O2 -ffp-contract=off for every x86-64: its arithmetic must not depend on
    # the machine, its speed undersells the CPU). Falls back to the parity build, and says so, when the host cannot compile.
    flags = "-O2 -ffp-contract=off (the parity build: this host could not build the timing-only object)"
    try:
        import subprocess
        import tempfile
        fast = os.path.join(tempfile.mkdtemp(prefix="jxlo_fast_"), "libjxlo_fast.so")
        fast_flags = ["-O3", "-march=native", "-std=c++17", "-fPIC", "-fopenmp", "-Wno-unused-function"]
        subprocess.run(["g++"] + fast_flags + ["-shared", "-o", fast, os.path.join(ROOT, "oracle", "jxlo_decoder.cc")], check=True,
                       capture_output=True, timeout=300)
        jxlo.LIB_PATH, jxlo._lib = fast, None
        flags = " ".join(fast_flags[:2]) + " (timing-only object built on this host; the parity build is -O2 -ffp-contract=off)"
    except Exception:  # noqa: BLE001
        pass
    threads = jxlo.lib().jxlo_set_threads(min(avail, 64))
    times, used = [], []
    t_start = time.time()
    while len(times) < 5 and (not times or time.time() - t_start + times[-1] < budget_s):
        t0, c0 = time.time(), time.process_time()
        d = jxlo.Decoded(data, dumps=False)
        times.append(time.time() - t0)
        used.append((time.process_time() - c0) / max(times[-1], 1e-9))
        d.close()
    best = min(times)
    # cores = what the threads really got (a cgroup CPU share can be far below the visible CPU count): CPU time / wall time
    cores = max(1, int(round(used[times.index(best)])))
    return {"value": round(xsize * ysize * 1e-6 / best, 3), "unit": "MP/s", "cores": cores, "threads": threads, "kind": "port", "build": flags,
            "sample": "%d full %dx%d frame decode(s) of the benchmark stream, best of %d" % (len(times), xsize, ysize, len(times))}


def end_to_end(J, datas, nframes, device, xsize, ysize, chunk=256):
    """End-to-end rate over `nframes` frames: from compressed bytes in host memory to RGB8 in (pinned) host memory, the span
    djxl times (tools/djxl_main.cc:415-422, tools/speed_stats.cc:107). Host parse (headers, DC groups, tables: libjxl_amd's
    host front-end), upload, the three GPU stages and the download of the pixels run as a pipeline over host threads, in
    chunks of `chunk` frames (an entropy launch lasts as long as its slowest section whatever the number of frames, so
    small chunks would only measure that latency): parser pool -> uploaders -> GPU stages -> downloaders, three context
    sets rotating. Not `value`: PCIe and host bound."""
    import concurrent.futures
    import queue
    import threading
    import torch
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    chunk = int(os.environ.get("JXLAMD_E2E_CHUNK", chunk))  # (measurement aid)
    chunk = max(1, min(chunk, nframes))
    # CPUs this process really gets (a container's CPU share does not show in the affinity mask): CPU time / wall time of
    # 64 frame parses on 64 threads. The host front-end is the end-to-end bottleneck, so the pools are sized by it.
    with concurrent.futures.ThreadPoolExecutor(64) as cal:
        w0, c0 = time.perf_counter(), time.process_time()
        for fr in list(cal.map(lambda i: J.Frame(datas[i % len(datas)], 1), range(128))):
            fr.close()
        ncpu_eff = max(1.0, min(float(ncpu), (time.process_time() - c0) / (time.perf_counter() - w0)))
    parse_threads = 1  # one frame per thread: frames in parallel, not DC groups in parallel
    # (the calibration is noisy on a shared box: 8.4, 6.0 and 5.7 for the same share in three runs. Too few parser threads
    # starve the pipeline, a few too many only time-slice: never fewer than 8.)
    parsers = max(8, min(64, int(round(ncpu_eff))))
    movers = 4 if ncpu_eff >= 8 else (2 if ncpu_eff >= 4 else 1)  # threads of the upload pool and of the download pool
    movers = int(os.environ.get("JXLAMD_E2E_MOVERS", movers))  # (measurement aid)
    parsers = int(os.environ.get("JXLAMD_E2E_PARSERS", parsers))
    nsets = 4  # one being uploaded, two on the GPU (the entropy launch of one beside the transform / filter of the other), one downloading
    sets = [[J.HipContext(device) for _ in range(chunk)] for _ in range(nsets)]
    try:
        sets[0][0].set_option("blocking_sync", 1)  # waiting threads sleep: the CPUs are needed by the parsers
    except J.JxlAmdError:
        pass
    free_sets = queue.Queue()
    for cs in sets:
        free_sets.put(cs)
    q_up, q_down = queue.Queue(maxsize=2), queue.Queue(maxsize=2)
    # pinned (page-locked) host buffers for the pixels, one per context
    pinned = [torch.empty((ysize, xsize, 3), dtype=torch.uint8, pin_memory=True) for _ in range(chunk * nsets)]
    outs = {id(c): pinned[i * chunk + j].numpy() for i, cs in enumerate(sets) for j, c in enumerate(cs)}
    errors = []
    done = [0]
    busy = {"wait_parse": 0.0, "wait_free_set": 0.0, "upload": 0.0, "gpu_stages": 0.0, "gpu_enqueue": 0.0, "enq_entropy": 0.0, "enq_transform": 0.0, "download": 0.0}
    kernel_ms = [0.0, 0.0, 0.0]

    def uploader(futs, pool):
        try:
            for c0 in range(0, nframes, chunk):
                t_a = time.perf_counter()
                fr = [f.result() for f in futs[c0:c0 + chunk]]
                t_b = time.perf_counter()
                cs = free_sets.get()
                t_c = time.perf_counter()
                list(pool.map(lambda cf: cf[0].upload(cf[1]), zip(cs, fr)))
                busy["wait_parse"] += t_b - t_a
                busy["wait_free_set"] += t_c - t_b
                busy["upload"] += time.perf_counter() - t_c
                q_up.put((cs, fr))
        except Exception as e:  # noqa: BLE001 (reported by the caller)
            errors.append(e)
        q_up.put(None)

    runners_left = [2]
    lock = threading.Lock()

    def runner():
        try:
            while True:
                item = q_up.get()
                if item is None:
                    q_up.put(None)  # (for the other runner)
                    break
                cs, fr = item
                live = cs[:len(fr)]
                t_a = time.perf_counter()
                J.run_entropy_batch(live)
                t_e = time.perf_counter()
                J.run_transform_batch(live)
                t_t = time.perf_counter()
                J.run_filter_color_batch(live)
                t_b = time.perf_counter()
                live[0].sync()  # (the set's stream: the downloads start on finished pixels)
                with lock:
                    busy["gpu_stages"] += time.perf_counter() - t_a
                    busy["gpu_enqueue"] += t_b - t_a
                    busy["enq_entropy"] += t_e - t_a
                    busy["enq_transform"] += t_t - t_e
                    for k in range(3):
                        kernel_ms[k] += live[0].stage_ms(k)
                q_down.put((cs, fr))
        except Exception as e:  # noqa: BLE001
            errors.append(e)
        with lock:
            runners_left[0] -= 1
            last = runners_left[0] == 0
        if last:
            q_down.put(None)

    def fetch(cf):
        c, f = cf
        J._check(J.lib().jxlhip_download_rgb8(c._h, outs[id(c)].ctypes.data, xsize * 3), "jxlhip_download_rgb8")
        f.close()

    def downloader(pool):
        try:
            while True:
                item = q_down.get()
                if item is None:
                    break
                cs, fr = item
                t_a = time.perf_counter()
                list(pool.map(fetch, zip(cs, fr)))
                busy["download"] += time.perf_counter() - t_a
                done[0] += len(fr)
                free_sets.put(cs)
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    # warm-up: one small chunk through everything (allocations, first touch of the output pages)
    warm = [J.Frame(datas[i % len(datas)], parse_threads) for i in range(min(chunk, 8))]
    for cs in sets:
        for i, c in enumerate(cs):  # every context allocates its device buffers now, not inside the timed region
            c.upload(warm[i % len(warm)])
        # the whole set once: the batch descriptions of its first context get their final size here (growing them later
        # means hipFree, which waits for everything the device has in flight)
        J.run_entropy_batch(cs); J.run_transform_batch(cs); J.run_filter_color_batch(cs)
        for c in cs[:len(warm)]:
            J._check(J.lib().jxlhip_download_rgb8(c._h, outs[id(c)].ctypes.data, xsize * 3), "jxlhip_download_rgb8")
    for f in warm:
        f.close()
    t0 = time.perf_counter()
    with concurrent.futures.ThreadPoolExecutor(parsers) as pool, concurrent.futures.ThreadPoolExecutor(movers) as up_pool, \
            concurrent.futures.ThreadPoolExecutor(movers) as down_pool:
        futs = [pool.submit(J.Frame, datas[i % len(datas)], parse_threads) for i in range(nframes)]
        ths = [threading.Thread(target=uploader, args=(futs, up_pool)), threading.Thread(target=runner), threading.Thread(target=runner),
               threading.Thread(target=downloader, args=(down_pool,))]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
    elapsed = time.perf_counter() - t0
    for cs in sets:
        for c in cs:
            c.close()
    if errors:
        raise errors[0]
    assert done[0] == nframes
    return {"value": round(nframes * xsize * ysize * 1e-6 / elapsed, 1), "unit": "MP/s", "frames": nframes,
            "ms_per_frame": round(elapsed / nframes * 1e3, 3), "host_cores": round(ncpu_eff, 1), "host_cores_visible": ncpu,
            "stage_busy_ms_per_frame": {k: round(v / nframes * 1e3, 3) for k, v in busy.items()},
            "kernel_ms_per_frame": {"entropy": round(kernel_ms[0] / nframes, 3), "transform": round(kernel_ms[1] / nframes, 3),
                                    "filter+colour": round(kernel_ms[2] / nframes, 3)},
            "host_threads": {"parse": "%d frames x %d threads" % (parsers, parse_threads), "upload": movers, "gpu_launch": 2, "download": movers},
            "span": "compressed bytes in host memory -> RGB8 in pinned host memory (tools/djxl_main.cc:415-422), host parse | H2D | entropy, "
                    "transform, filter+colour in chunks of %d frames, two chunks on the GPU at a time | D2H pipelined over host threads; "
                    "stage_busy_ms_per_frame = wall time of each pipeline thread (gpu_stages: summed over its two threads)" % chunk}


def single_image_api(J, data, xsize, ysize, reps=10, channels=3):
    """One image through the drop-in boundary, the way djxl times it (tools/djxl_main.cc:415-422: wall clock around the whole
    DecodeImageJXL call sequence, compressed bytes in memory -> RGB8 in the caller's buffer; tools/speed_stats.cc:23-69: with
    three or more repetitions the geometric mean without the first)."""
    import ctypes
    import math
    L = J.lib()
    vp = ctypes.c_void_p
    L.JxlDecoderCreate.restype = vp
    L.JxlDecoderCreate.argtypes = [vp]
    L.JxlThreadParallelRunnerCreate.restype = vp
    L.JxlThreadParallelRunnerCreate.argtypes = [vp, ctypes.c_size_t]
    L.JxlThreadParallelRunnerDestroy.argtypes = [vp]
    for fn, args in (("JxlDecoderDestroy", [vp]), ("JxlDecoderSubscribeEvents", [vp, ctypes.c_int]),
                     ("JxlDecoderSetInput", [vp, ctypes.c_char_p, ctypes.c_size_t]), ("JxlDecoderCloseInput", [vp]),
                     ("JxlDecoderProcessInput", [vp]), ("JxlDecoderSetParallelRunner", [vp, vp, vp]),
                     ("JxlDecoderSetImageOutBuffer", [vp, vp, vp, ctypes.c_size_t])):
        getattr(L, fn).argtypes = args

    class Fmt(ctypes.Structure):
        _fields_ = [("num_channels", ctypes.c_uint32), ("data_type", ctypes.c_int), ("endianness", ctypes.c_int), ("align", ctypes.c_size_t)]
    fmt = Fmt(channels, 2, 0, 0)  # RGB (or RGBA), JXL_TYPE_UINT8, native endian
    out = ctypes.create_string_buffer(xsize * ysize * channels)
    try:
        threads = len(os.sched_getaffinity(0))
    except AttributeError:
        threads = os.cpu_count() or 1
    threads = max(1, min(threads, 32))
    pool = L.JxlThreadParallelRunnerCreate(None, threads)
    runner = ctypes.cast(L.JxlThreadParallelRunner, vp)
    times = []
    for _ in range(reps):
        t0 = time.perf_counter()
        dec = L.JxlDecoderCreate(None)
        L.JxlDecoderSetParallelRunner(dec, runner, pool)
        L.JxlDecoderSubscribeEvents(dec, 0x1000)  # JXL_DEC_FULL_IMAGE
        L.JxlDecoderSetInput(dec, data, len(data))
        L.JxlDecoderCloseInput(dec)
        status = L.JxlDecoderProcessInput(dec)
        if status == 5:  # JXL_DEC_NEED_IMAGE_OUT_BUFFER
            L.JxlDecoderSetImageOutBuffer(dec, ctypes.byref(fmt), out, len(out))
            status = L.JxlDecoderProcessInput(dec)
        L.JxlDecoderDestroy(dec)
        times.append(time.perf_counter() - t0)
        if status != 0x1000:
            L.JxlThreadParallelRunnerDestroy(pool)
            return {"error": "JxlDecoderProcessInput returned %d" % status}
    L.JxlThreadParallelRunnerDestroy(pool)
    tail = times[1:] if len(times) >= 3 else times[-1:]
    gm = math.exp(sum(math.log(t) for t in tail) / len(tail))
    return {"value": round(xsize * ysize * 1e-6 / gm, 1), "unit": "MP/s", "seconds": round(gm, 5), "reps": reps, "runner_threads": threads,
            "span": "JxlDecoderCreate .. JXL_DEC_FULL_IMAGE of ONE %dx%d frame through the JxlDecoder C API with JxlThreadParallelRunner "
                    "(tools/djxl_main.cc:415-422), geometric mean without the first repetition (tools/speed_stats.cc:23-69)" % (xsize, ysize)}


def system_libjxl_baseline(data, xsize, ysize, threads):
    """If the box has a libjxl of its own (probe, never assume), times it on the same stream through the
    same JxlDecoder C API with its thread pool and returns a cpu_baseline dict (kind "reference"), else None."""
    import ctypes
    import ctypes.util
    name, tname = ctypes.util.find_library("jxl"), ctypes.util.find_library("jxl_threads")
    if not name or not tname:
        return None
    try:
        L, T = ctypes.CDLL(name), ctypes.CDLL(tname)
        vp = ctypes.c_void_p
        L.JxlDecoderCreate.restype = vp
        L.JxlDecoderCreate.argtypes = [vp]
        T.JxlThreadParallelRunnerCreate.restype = vp
        T.JxlThreadParallelRunnerCreate.argtypes = [vp, ctypes.c_size_t]
        for fn, args in (("JxlDecoderDestroy", [vp]), ("JxlDecoderSubscribeEvents", [vp, ctypes.c_int]),
                         ("JxlDecoderSetInput", [vp, ctypes.c_char_p, ctypes.c_size_t]), ("JxlDecoderCloseInput", [vp]),
                         ("JxlDecoderProcessInput", [vp]), ("JxlDecoderSetParallelRunner", [vp, vp, vp]),
                         ("JxlDecoderSetImageOutBuffer", [vp, vp, vp, ctypes.c_size_t])):
            getattr(L, fn).argtypes = args

        class Fmt(ctypes.Structure):
            _fields_ = [("num_channels", ctypes.c_uint32), ("data_type", ctypes.c_int), ("endianness", ctypes.c_int),
                        ("align", ctypes.c_size_t)]
        fmt = Fmt(3, 2, 0, 0)  # RGB, JXL_TYPE_UINT8, native endian
        out = ctypes.create_string_buffer(xsize * ysize * 3)
        pool = T.JxlThreadParallelRunnerCreate(None, threads)
        runner = ctypes.cast(T.JxlThreadParallelRunner, vp)
        times = []
        for _ in range(4):
            t0 = time.time()
            dec = L.JxlDecoderCreate(None)
            L.JxlDecoderSetParallelRunner(dec, runner, pool)
            L.JxlDecoderSubscribeEvents(dec, 0x1000)  # JXL_DEC_FULL_IMAGE
            L.JxlDecoderSetInput(dec, data, len(data))
            L.JxlDecoderCloseInput(dec)
            status = L.JxlDecoderProcessInput(dec)
            if status == 5:  # JXL_DEC_NEED_IMAGE_OUT_BUFFER
                L.JxlDecoderSetImageOutBuffer(dec, ctypes.byref(fmt), out, len(out))
                status = L.JxlDecoderProcessInput(dec)
            L.JxlDecoderDestroy(dec)
            if status != 0x1000:
                return None
            times.append(time.time() - t0)
        return {"value": round(xsize * ysize * 1e-6 / min(times), 3), "unit": "MP/s", "cores": threads, "kind": "reference",
                "sample": "system %s, %d full %dx%d frame decodes of the benchmark stream, best of %d" % (name, len(times), xsize, ysize, len(times))}
    except (OSError, AttributeError):
        return None


def lossless_main(args, J, sharding, torch, dist, rank, local_rank, world, xsize, ysize):
    """BASELINE.json configs[3]: Modular lossless decode. A step decodes `--batch` frames (every stream of every frame as one
    lane of ONE k_modular_streams launch, then the inverse transforms and the sample conversion per frame)."""
    import numpy as np
    # (the VarDCT default of 640 would not fit: ~0.35 GB of channel buffers per frame; 384 frames = 134 of the 288 GB, and
    # the launch time is set by the longest stream, so frames per launch is what throughput follows: DESIGN.md §8.5)
    batch = args.batch if args.batch != 640 else 384
    ndistinct = max(1, min(args.distinct, batch))
    flags = J.LOSSLESS_RCT | J.LOSSLESS_SQUEEZE | J.LOSSLESS_WP
    if args.lossless_flags >= 0:  # measurement aid: other feature sets of the synthetic encoder
        flags = args.lossless_flags
    datas = []
    for i in range(ndistinct):
        cache = "/tmp/libjxl_amd_bench_lossless_%dx%d_s%d_f%d.jxl" % (xsize, ysize, 177 + i, flags)
        if os.path.exists(cache):
            datas.append(open(cache, "rb").read())
            continue
        d = J.encode_lossless(J.synth_image(xsize, ysize, 177 + i), flags)
        try:
            open(cache + ".%d" % os.getpid(), "wb").write(d)
            os.replace(cache + ".%d" % os.getpid(), cache)
        except OSError:
            pass
        datas.append(d)
    frames = [J.ModFrame(d) for d in datas]
    ctxs = [J.HipContext(local_rank) for _ in range(batch)]
    for i, c in enumerate(ctxs):
        c.upload_modular(frames[i % ndistinct])

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(1, args.warmup)):
        J.run_modular_batch(ctxs)
    r, status, _ = ctxs[0].modular_status()
    if r:
        raise SystemExit("corrupt Modular streams: %r" % status)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        J.run_modular_batch(ctxs)
    barrier()
    elapsed = time.perf_counter() - t0
    total_frames, max_elapsed = sharding.aggregate(batch * args.steps, elapsed, dist)
    launch_ms = ctxs[0].stage_ms(0)
    if rank == 0:
        px = xsize * ysize
        comp = sum(len(d) for d in datas) / float(ndistinct)
        streams = sum(f.info["num_streams"] for f in frames) / float(ndistinct)
        # algorithmic bytes per frame: compressed bytes read + every decoded sample written once as int32 (3 channels) by the
        # stream kernel; the inverse transforms and the conversion move another ~(2 * 12 + 12 + 3) B/px in their own launches
        alg = comp + 12.0 * px
        out = {"metric": "megapixels/sec decode, %dx%d Modular lossless (Squeeze + MA tree)" % (xsize, ysize),
               "value": round(total_frames * px * 1e-6 / max_elapsed, 2), "unit": "MP/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(max_elapsed / args.steps * 1e3, 3), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
               "config": {"workload": "%dx%d RGB8 Modular lossless decode (RCT + Squeeze + MA tree with weighted predictor, rANS), %d "
                                      "frames/step/GPU, inputs resident in HBM" % (xsize, ysize, batch),
                          "bpp": round(comp * 8.0 / px, 3), "streams_per_frame": streams, "distinct_frames": ndistinct,
                          "parallelism": "frames sharded over %d GPU(s), no data-path collective" % world},
               "roofline": {"bound": "hbm", "kernel": "k_modular_streams + inverse transforms + output (one timed span)",
                            "achieved": round(alg * batch / (launch_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(alg * batch / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": None,
                            "frames_per_launch": batch, "launch_ms": round(launch_ms, 3),
                            "note": "a Modular stream is a pixel-serial adaptive decode (one lane per stream): latency-bound, not HBM-bound"}}
        if not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import jxlo
            t = []
            for _ in range(2):
                t1 = time.time()
                jxlo.Decoded(datas[0], dumps=False).close()
                t.append(time.time() - t1)
            out["cpu_baseline"] = {"value": round(px * 1e-6 / min(t), 3), "unit": "MP/s", "cores": 1, "kind": "port",
                                   "sample": "2 full frame decodes of the benchmark stream by the oracle (scalar, one thread), best"}
        print(json.dumps(out))
    for c in ctxs:
        c.close()
    for f in frames:
        f.close()
    if dist is not None:
        dist.destroy_process_group()


def encode_main(args, J, sharding, torch, dist, rank, local_rank, world, xsize, ysize):
    """BASELINE.json configs[4] (VarDCT encode), first slice: the forward path on the GPU (colour, sharpening, transform
    selection, forward DCT, quantisation: jxlhip_enc_forward). A step runs the kernel sequence over `--batch` frames with
    the RGB8 input resident in HBM; entropy coding stays on the host and is reported in `e2e`, not in `value`."""
    batch = args.batch if args.batch != 640 else 64
    img = J.synth_image(xsize, ysize, 177)
    ctx = J.HipContext(local_rank)
    t = {}
    data = J.encode_rgb8_gpu(img, ctx, timings=t, distance=args.distance, cfl_fit=1)  # (leaves the image resident on the device)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(1, args.warmup)):
        ctx.enc_rerun(batch)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        span_ms, transform_ms = ctx.enc_rerun(batch)
    barrier()
    elapsed = time.perf_counter() - t0
    total_frames, max_elapsed = sharding.aggregate(batch * args.steps, elapsed, dist)
    if rank == 0:
        px = xsize * ysize
        # forward transform kernel: reads the three f32 planes once, writes int32 coefficients once
        alg = 24.0 * px
        out = {"metric": "megapixels/sec VarDCT encode forward path, %dx%d d%.1f" % (xsize, ysize, args.distance),
               "value": round(total_frames * px * 1e-6 / max_elapsed, 2), "unit": "MP/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(max_elapsed / args.steps * 1e3, 3), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "%dx%d RGB8 -> XYB, sharpening, transform selection, forward DCT 8..64, per-tile chroma-from-luma fit, "
                                      "quantisation (pixel-domain half of a VarDCT encode), %d frames/step/GPU, input resident in "
                                      "HBM; entropy coding on the host is outside `value` (see e2e)" % (xsize, ysize, batch),
                          "bytes_out": len(data), "bpp": round(len(data) * 8.0 / px, 3),
                          "parallelism": "frames sharded over %d GPU(s), no data-path collective" % world},
               "roofline": {"bound": "hbm", "kernel": "k_enc_transform_tile", "achieved": round(alg / (transform_ms * 1e-3) / 1e9, 2),
                            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg / (transform_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                            "traffic": None, "launch_ms": round(transform_ms, 4), "algorithmic_bytes_per_launch": alg,
                            "kernels_ms_per_frame": round(span_ms / batch, 4)}}
        if world == 1:
            def best_of(device_tokens):
                best = None
                for _ in range(3):
                    t1 = time.perf_counter()
                    d = J.encode_rgb8_gpu(img, ctx, timings=t, device_tokens=device_tokens, distance=args.distance, cfl_fit=1)
                    dt = time.perf_counter() - t1
                    if best is None or dt < best[0]:
                        best = (dt, dict(t), d)
                return best
            host_tok, best = best_of(False), best_of(True)
            assert best[2] == host_tok[2] == data  # the same codestream whichever side tokenises
            out["e2e"] = {"value": round(px * 1e-6 / best[0], 2), "unit": "MP/s", "seconds_per_frame": round(best[0], 4),
                          "forward_call_s": round(best[1]["forward_s"], 4), "host_entropy_coding_s": round(best[1]["assemble_s"], 4),
                          "device_tokens": best[1]["device_tokens"],
                          "host_tokenised": {"seconds_per_frame": round(host_tok[0], 4), "forward_call_s": round(host_tok[1]["forward_s"], 4),
                                             "host_entropy_coding_s": round(host_tok[1]["assemble_s"], 4)},
                          "note": "one frame, RGB8 in host memory to codestream bytes: upload, kernels (AC tokenisation included), "
                                  "token download, host histogram clustering + rANS coding; host_tokenised = the coefficients "
                                  "copied back and tokenised by the host instead (the same bytes)"}
            if not args.no_cpu_baseline:
                t1, c1 = time.perf_counter(), time.process_time()
                J.enc_forward_model(img, None, distance=args.distance, cfl_fit=1)
                dt = time.perf_counter() - t1
                cores = max(1, int(round((time.process_time() - c1) / dt)))  # the cores the OpenMP loops really got (cgroup share)
                out["cpu_baseline"] = {"value": round(px * 1e-6 / dt, 3), "unit": "MP/s", "cores": cores, "kind": "port",
                                       "sample": "the same forward path of one frame by the CPU stream writer (OpenMP over groups)"}
        print(json.dumps(out))
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=640, help="frames per step per GPU")
    ap.add_argument("--size", default="3840x2160")
    ap.add_argument("--distance", type=float, default=1.0)
    ap.add_argument("--shard", choices=("frames", "bands"), default="frames",
                    help="multi-GPU split: whole frames per rank (weak scaling) or, for very large frames, one band of rows "
                         "of 256x256 groups of every frame per rank (strong scaling, no exchange between ranks)")
    ap.add_argument("--halo", action="store_true",
                    help="--shard bands: every rank decodes ONLY its own rows of groups and the few rows its filters read beyond "
                         "them are exchanged with the neighbouring ranks after the transform stage (RCCL send / recv of dense "
                         "device blocks; jxlhip_halo_*), instead of decoding one group row of overlap either side")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-libjxl-tables", action="store_true",
                    help="skip the second measurement of the default line (the same schedule over streams with 128 histogram clusters)")
    ap.add_argument("--no-pipeline", action="store_true", help="one frame set: entropy, then transform+filter, in sequence")
    ap.add_argument("--two-set", action="store_true",
                    help="the round-1 schedule: entropy(set A) beside transform(B) -> filter+colour(B), the sets' XYB planes shared")
    ap.add_argument("--three-stage", action="store_true",
                    help="two frame sets, three concurrent launches per step: entropy(A) | transform(B) | filter+colour(A, planes of "
                         "the previous step; option filter_async), instead of entropy(A) | transform(B) -> filter(B)")
    ap.add_argument("--chain", action="store_true",
                    help="one frame set, filter + colour of step k on a second stream under the entropy launch of step k + 1 "
                         "(option filter_async), instead of the default software pipeline over two frame sets")
    ap.add_argument("--no-share-planes", action="store_true", help="every context of both pipelined sets keeps its own XYB planes")
    ap.add_argument("--sets", type=int, default=0,
                    help="S > 0: the batch is split into S frame sets, each with its own stream running entropy -> transform -> "
                         "filter+colour for its frames every step, all S streams free-running beside each other (no set waits for "
                         "another: the long tail of one set's entropy launch -- its slowest section -- overlaps the other sets' "
                         "launches). 0 = the two-set schedule selected by the other flags")
    ap.add_argument("--distinct", type=int, default=8, help="distinct synthetic frames (seeds 177, 178, ...) cycled through every frame set")
    ap.add_argument("--e2e-frames", type=int, default=2048,
                    help="frames of the end-to-end measurement (compressed bytes in host memory -> RGB8 in host memory, host parse / "
                         "upload / GPU stages / download pipelined over host threads); 0 = skip")
    ap.add_argument("--max-clusters", type=int, default=0, help="sensitivity runs: histogram clusters of the synthetic encoder (0 = its default 64)")
    ap.add_argument("--workload", choices=("vardct", "lossless", "encode"), default="vardct",
                    help="vardct: BASELINE.json configs[1] (the headline); lossless: configs[3], 3840x2160 Modular lossless "
                         "(Squeeze + MA tree + weighted predictor) through k_modular_streams")
    ap.add_argument("--ac-code-mode", type=int, default=0,
                    help="sensitivity runs: AC coefficient streams with prefix codes (1: what libjxl's fastest efforts emit), LZ77 (2) or both (3)")
    ap.add_argument("--lossless-flags", type=int, default=-1,
                    help="--workload lossless: feature bits of the synthetic encoder (libjxl_amd.LOSSLESS_*) instead of RCT + Squeeze + WP")
    ap.add_argument("--launch-check", action="store_true",
                    help="only check the multi-rank launch (gloo, no GPU needed): every rank reports, rank 0 prints the ranks it saw")
    args = ap.parse_args()
    xsize, ysize = [int(v) for v in args.size.split("x")]

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # Not started by torch.distributed.run: start the N ranks ourselves, as a CHILD process and before anything here has
        # touched the GPU (a process that has initialised HIP must never exec another program), and pass its exit code on.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE); refusing to report a wrong n_gpus" % (args.gpus, world))
    dist = None
    import torch
    if args.launch_check:
        import torch.distributed as dist
        dist.init_process_group("gloo")
        seen = [None] * world
        dist.all_gather_object(seen, {"rank": rank, "local_rank": local_rank, "pid": os.getpid()})
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world, "ranks": sorted(s["rank"] for s in seen),
                              "distinct_processes": len({s["pid"] for s in seen})}))
        dist.destroy_process_group()
        return
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    import libjxl_amd as J
    from libjxl_amd import sharding
    J.lib()  # fails loudly if the HIP extension is missing
    if args.workload == "encode":
        return encode_main(args, J, sharding, torch, dist, rank, local_rank, world, xsize, ysize)
    if args.workload == "lossless":
        return lossless_main(args, J, sharding, torch, dist, rank, local_rank, world, xsize, ysize)

    # Frames larger than 4K without an explicit --batch: as many per step as two frame sets of ~21 B per pixel (int16
    # coefficients, f32 XYB planes, RGB8) leave room for in HBM: 16 for 16384 x 16384 (63 GP/s; 34 at 8 frames per step: a
    # launch lasts as long as its longest section whatever the number of frames), and one distinct frame (encoding a 16K
    # frame on the host takes a minute).
    if args.batch == 640 and xsize * ysize > 2 * 3840 * 2160:
        args.batch = max(1, min(640, int(185e9 / (2 * 21.0 * xsize * ysize))))
        args.distinct = min(args.distinct, 1)
    # `--distinct` different frames (the synthetic image generator with seeds 177, 178, ...), cycled through every frame
    # set: sections, token counts and entropy tables differ from frame to frame, so the workgroups of a launch do not all
    # finish together
    ndistinct = max(1, min(args.distinct, args.batch))
    datas = [make_stream(xsize, ysize, args.distance, 177 + i, args.max_clusters, args.ac_code_mode) for i in range(ndistinct)]
    frames = [J.Frame(d, threads=min(8, os.cpu_count() or 1)) for d in datas]
    data, frame = datas[0], frames[0]
    info = dict(frame.info)
    info["ac_bytes"] = sum(f.info["ac_bytes"] for f in frames) / float(ndistinct)  # mean over the distinct frames
    # LDS of one entropy workgroup (one frame: jxl_hip_entropy_lanes.h LanesLdsLayout): alias tables, context map,
    # nnz / config tables, 64 lanes of rings and line buffers; a CU has 160 KB
    tables = []
    for f in frames:
        lds = (f.info["num_clusters"] << f.info["log_alpha"]) * 8 + ((f.info["ctx_map_size"] + 16 + 15) & ~15) + 128 + 128 + 64 * 200
        tables.append({"clusters": f.info["num_clusters"], "log_alpha": f.info["log_alpha"], "lds_bytes": lds,
                       "workgroups_per_cu": (160 * 1024) // lds})
    # Two sets of `batch` frames: while one set is in the (latency-bound, serial per section) entropy stage, the other
    # set's coefficients go through the (bandwidth-bound) transform + filter + colour stages. Every step runs every
    # stage once over `batch` frames, so a step completes `batch` frames; a frame's latency is two steps.
    free_running = args.sets > 0
    nsets = args.sets if free_running else (1 if (args.no_pipeline or args.chain) else 2)
    chain = args.chain and not args.no_pipeline and not free_running
    if free_running:
        per_set = [args.batch // nsets + (1 if i < args.batch % nsets else 0) for i in range(nsets)]
        sets = [[J.HipContext(local_rank) for _ in range(n)] for n in per_set if n]
        nsets = len(sets)
    else:
        sets = [[J.HipContext(local_rank) for _ in range(args.batch)] for _ in range(nsets)]
    band = None
    share = 1.0  # fraction of every frame's pixels this rank produces
    if args.shard == "bands":
        group_rows = (ysize + 255) // 256
        band = sharding.band_of(group_rows, rank, world)
        if band[0] == band[1]:
            raise SystemExit("more ranks than rows of groups: use --shard frames")
        share = (min(band[1] * 256, ysize) - band[0] * 256) / float(ysize)
    # default: three concurrent launches per step (entropy(A) | transform(B) | filter+colour(A, previous step)): with frames
    # that differ, an entropy launch ends with a long tail (its slowest section); the third stream keeps the machine busy
    three = not args.two_set and nsets == 2 and not free_running
    if three:
        for cs in sets:
            cs[0].set_option("filter_async", 1)
    if nsets == 2 and not free_running:
        # this schedule keeps a set's entropy launch and the other set's transform / filter launches in flight together
        # and enqueues every launch itself, in an order in which each one can start: it asks for the entropy gate (off by
        # default in the library; ignored by it when the runtime may run kernels one at a time)
        for cs in sets:
            cs[0].set_option("entropy_gate", 1)
    if nsets == 2 and not args.no_share_planes and not three and not free_running:  # (three-stage: both sets' planes are live at once)
        # the XYB planes of a frame only live between its transform and its filter stage, and the two sets are never in
        # those stages at the same time: set 1 keeps its planes in set 0's buffers (100 MB less per pair of 4K frames)
        for a, b in zip(sets[0], sets[1]):
            b.share_planes(a)
    if chain:
        sets[0][0].set_option("filter_async", 1)
    def load(frs):
        nth = 0
        for cs in sets:
            for c in cs:
                if band is not None and args.halo:
                    c.set_option("band_halo", 1)
                c.upload(frs[nth % len(frs)], band=band)
                nth += 1

    load(frames)

    def exchange_halos(cs):
        # after the transform stage of a band set: this rank's boundary rows to its neighbours, theirs beside its band; one
        # block per frame and side, all frames of the set in one tensor per message
        if band is None or not args.halo or world == 1:
            return
        n = cs[0].halo_floats()

        # Ordering: the library copies on ITS stream, the transport (c10d over RCCL) orders its
        # communication stream against torch's CURRENT stream only. The _batch calls tie the two together with events:
        # pack -> torch's stream waits for the copies -> dist.send; dist.recv (the current stream waits for the receive
        # when the call returns) -> the library's stream waits for torch's stream -> unpack -> the set's filter launch.
        # No host synchronisation anywhere.
        ts = torch.cuda.current_stream().cuda_stream

        def pack(side):
            t = torch.empty(len(cs) * n, dtype=torch.float32, device="cuda")
            J.halo_pack_batch(cs, side, t.data_ptr(), n * 4, ts)
            return t

        def unpack(side, t):
            J.halo_unpack_batch(cs, side, t.data_ptr(), n * 4, ts)

        def recv(peer):
            t = torch.empty(len(cs) * n, dtype=torch.float32, device="cuda")
            dist.recv(t, peer)
            return t

        sharding.exchange_halos(rank, world, pack, unpack, lambda t, peer: dist.send(t, peer), recv)
    def prime():  # every set holds decoded coefficients before the first (warmup) step
        for cs in sets:
            J.run_entropy_batch(cs)
        for cs in sets:
            for c in cs:
                c.sync()

    prime()
    step_no = [0]

    def step():
        k = step_no[0]
        step_no[0] += 1
        if free_running:
            # every set: its three stages, in order, on its own stream; the host only enqueues, the streams drift apart by
            # themselves (whoever gets resources first), so one set's latency-bound entropy launch runs beside other sets'
            # transform / filter launches and beside other entropy launches
            for cs in sets:
                J.run_entropy_batch(cs)
                J.run_transform_batch(cs)
                J.run_filter_color_batch(cs)
            return sets[0]
        ent = sets[k % nsets]
        down = sets[(k + 1) % nsets]
        if three:
            # three independent launches: the planes `ent` got from its transform in the previous step are filtered (on its
            # second stream) while its next coefficients are decoded and the other set is transformed
            # The entropy launch must find the chip's LDS free (EntropyGate in jxl_hip_api.hip): the transform launch of the
            # other set, enqueued behind it, is held back by the library until the entropy workgroups are resident. Same
            # box, same call (GP/s): filter, entropy, transform 70.4 / 70.0 with the gate, 62.4 / 69.6 without (two modes:
            # in the bad one the entropy launch takes 153 ms); entropy, filter, transform 65.6 / 65.0 (BENCH_ORDER=efd).
            if os.environ.get("BENCH_ORDER", "fed") == "efd":  # measurement aid
                J.run_entropy_batch(ent)
                J.run_filter_color_batch(ent)
            else:
                J.run_filter_color_batch(ent)
                J.run_entropy_batch(ent)
            J.run_transform_batch(down)
            exchange_halos(down)
            return ent
        J.run_entropy_batch(ent)  # one launch: the per-section decoders of all frames of the set share the GPU
        J.run_transform_batch(down)      # one launch per transform kernel for the whole set
        exchange_halos(down)
        J.run_filter_color_batch(down)   # one fused filter + colour launch
        # No host synchronisation inside a step: all work of a set is ordered on that set's first stream (entropy ->
        # transform -> filter -> next entropy ...), the two sets' streams overlap on the device, and the host only
        # enqueues. The timed region is closed by the barrier's device synchronisation.
        return ent

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    for cs in sets:
        cs[0].sync()
    for cs in sets:
        r, flags = cs[0].errors()
        if r:
            raise SystemExit("entropy kernel reported corrupt sections: %r" % flags)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    # duration of the entropy launches: HIP events on the stream each batch kernel was launched on; the events of the
    # last launch of every set are still in place after the region
    if free_running:  # frames-weighted mean of the sets' last entropy launches, expressed per whole batch like the other modes
        entropy_ms = sum(cs[0].stage_ms(0) for cs in sets) * args.steps
    else:
        entropy_ms = sum(cs[0].stage_ms(0) for cs in sets) / float(nsets) * args.steps
    frames_local = args.batch * args.steps * share
    total_frames, max_elapsed = sharding.aggregate(frames_local, elapsed, dist)
    # transform and filter+colour: the same batched launches over one set, run alone after the timed region (inside the
    # region their kernels share the GPU with the concurrently running entropy kernel); per frame = launch time / frames
    iso = [0.0, 0.0]
    c = sets[0][0]
    for which, fn in ((1, J.run_transform_batch), (2, J.run_filter_color_batch)):
        best = 1e9
        for _ in range(3):
            fn(sets[0])
            c.sync()
            best = min(best, c.stage_ms(which))
        iso[which - 1] = best / len(sets[0])
    # the entropy launch alone too (inside the region two of them overlap each other and the other stages)
    ent_alone = 1e9
    for _ in range(2):
        J.run_entropy_batch(sets[0])
        c.sync()
        ent_alone = min(ent_alone, c.stage_ms(0))
    stage_ms = [entropy_ms / args.steps / args.batch, iso[0], iso[1]]
    # The same schedule over streams with libjxl-sized entropy tables: the synthetic encoder clusters
    # to at most 64 histograms by default, libjxl to kClustersLimit = 128 (lib/jxl/enc_ans.cc:931), whose alias tables
    # (128 x 2^6 entries x 8 B = 64 KB per frame) no longer leave every frame of the launch resident in LDS: the lane kernel
    # then reads them in place from global memory. Reported beside `value`, never instead of it.
    libjxl_tables = None
    if world == 1 and args.max_clusters == 0 and args.ac_code_mode == 0 and not args.no_libjxl_tables and args.shard == "frames":
        datas128 = [make_stream(xsize, ysize, args.distance, 177 + i, 128, 0) for i in range(ndistinct)]
        frames128 = [J.Frame(d, threads=min(8, os.cpu_count() or 1)) for d in datas128]
        # Frames per step: 512 (when `value` runs more). 64 KB of alias tables in the 8-byte form leave one workgroup per CU
        # (86 KB per frame: 256 frames per launch, 45.8 GP/s); the six-byte form of jxl_hip_entropy_lanes.h (70 KB per frame) keeps
        # two per CU resident = 512 frames per launch. More frames than that in one launch fall back to the tables read in
        # place from global memory (the line's second figure, at `value`'s frames per step, for continuity with round 3).
        lds128 = max(128 + 16384 + 32768 + 256 + ((f.info["ctx_map_size"] + 16 + 15) & ~15) + 128 + 64 * 200 for f in frames128)
        full_sets = sets

        def run128(batch128):
            nonlocal sets
            sets = [cs[:batch128] for cs in full_sets]
            load(frames128)
            prime()
            for _ in range(max(1, args.warmup)):
                step()
            barrier()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step()
            barrier()
            el = time.perf_counter() - t1
            return round(batch128 * args.steps * xsize * ysize * 1e-6 / el, 2), round(el / args.steps * 1e3, 3)

        batch128 = min(512, args.batch)
        v128, ms128 = run128(batch128)
        libjxl_tables = {"value": v128, "unit": "MP/s", "ms_per_step": ms128, "frames_per_step": batch128, "max_clusters": 128,
                         "clusters": [f.info["num_clusters"] for f in frames128], "log_alpha": [f.info["log_alpha"] for f in frames128],
                         "lds_bytes_per_frame": lds128,
                         "bpp": round(sum(len(d) for d in datas128) / float(ndistinct) * 8.0 / (xsize * ysize), 3),
                         "note": "same frames, schedule and timed region as `value`; histograms clustered to libjxl's limit of 128 "
                                 "(enc_ans.cc:931) instead of the synthetic encoder's default 64. Their alias tables stay in LDS in a "
                                 "six-byte form (hand-written trip, three reads per token) at two frames per CU, hence 512 frames per step"}
        if args.batch > batch128:
            v2, ms2 = run128(args.batch)
            libjxl_tables["at_value_frames_per_step"] = {"value": v2, "ms_per_step": ms2, "frames_per_step": args.batch,
                                                         "note": "more frames than stay resident with the tables in LDS: read in place "
                                                                 "from global memory (the C++ trip), as in round 3"}
        sets = full_sets
        for f in frames128:
            f.close()
    for cs in sets:  # the frame sets' device memory is released before the end-to-end measurement allocates its own
        for c in cs:
            c.close()

    if rank == 0:
        px = xsize * ysize
        mps = total_frames * px * 1e-6 / max_elapsed
        bpp = sum(len(d) for d in datas) / float(ndistinct) * 8.0 / px
        # algorithmic bytes per launch (one frame) of each stage, SURVEY.md §8d / DESIGN.md:
        alg = {
            "entropy (k_entropy_lanes)": info["ac_bytes"] + 6.0 * px,          # bitstream read + int16 coefficients written
            "transform (k_idct_fast/k_dct/k_special)": (6.0 + 0.4 + 12.0) * px,  # coefficients + side info read, f32 XYB written
            "filter+colour (k_filter_rows2)": (12.0 + 0.06 + 3.0) * px,           # f32 XYB + sigma read, RGB8 written
        }
        names = list(alg)
        dom = max(range(3), key=lambda s: stage_ms[s])
        achieved = alg[names[dom]] / (stage_ms[dom] * 1e-3) / 1e9
        frames_per_launch = args.batch if dom == 0 else 1
        # HBM traffic of the dominant kernel: PMC counters from separate rocprofv3 passes of this command (committed
        # summary; 2 x FETCH_SIZE + WRITE_SIZE per dispatch), only quoted when taken at the same frames per launch
        traffic, traffic_source = None, None
        try:
            pmc_name = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_traffic.json"))[-1]  # the latest round's
            pmc = json.load(open(os.path.join(ROOT, "profiles", pmc_name)))
            key = ["k_entropy_lanes", "k_idct_fast<short, 4, 4>", "k_filter_rows2"][dom]
            # (several instantiations of a kernel may be in the summary -- the libjxl-tables leg runs others: the one this
            # command dispatched most often is the default line's)
            matches = sorted(((v.get("dispatches", 0), name) for name, v in pmc.items() if isinstance(v, dict) and key in name), reverse=True)[:1]
            for _, name in matches:
                v = pmc[name]
                if pmc.get("_frames_per_launch") == args.batch:
                    traffic = int(v["hbm_kib_per_dispatch"] * 1024)  # FETCH_SIZE doubled (gfx950: the guide's correction) + WRITE_SIZE
                    # (PMC counters cannot be collected inside this run: the figure is the committed profile's, named with the
                    # commit it was taken at, for a launch of the same frames per launch)
                    traffic_source = "profiles/%s (separate rocprofv3 --pmc passes of this command at commit %s)" % (pmc_name, pmc.get("_commit", "unknown"))
        except (OSError, ValueError, KeyError, TypeError):
            pass
        out = {
            "metric": "megapixels/sec decode, %dx%d VarDCT d%.1f" % (xsize, ysize, args.distance),
            "value": round(mps, 2),
            "unit": "MP/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(max_elapsed / args.steps * 1e3, 3),
            "value_libjxl_tables": libjxl_tables["value"] if libjxl_tables else None,
            "libjxl_tables": libjxl_tables,
            "higher_is_better": True,
            "scaling": "weak" if args.shard == "frames" else "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "%dx%d RGB8 VarDCT d%.1f decode (gab+EPF1, 1 pass), %d frames/step/GPU, inputs resident in HBM" % (
                xsize, ysize, args.distance, args.batch),
                "timed_region": "the GPU stages of the frame decode: AC entropy decode of every 256x256 group, dequantisation + chroma "
                                "from luma + inverse transforms, Gaborish + EPF + XYB->sRGB + RGB8 store; the compressed sections and "
                                "the per-frame tables (headers, TOC, DC image, quant field, histograms: host front-end) are resident in "
                                "HBM before it starts, the pixels stay in HBM. The rate from compressed bytes in host memory to pixels "
                                "in host memory (host parse, PCIe both ways: the reference's own metric) is `e2e`, never `value`",
                "bpp": round(bpp, 3), "groups_per_frame": info["num_groups"],
                "frames_per_step_per_gpu": args.batch, "distinct_frames": ndistinct, "entropy_tables": tables, "pipeline": ("%d frame sets, each on its own stream (entropy -> transform -> filter+colour every step), free-running" % nsets) if free_running else ("2 frame sets, 3 concurrent launches: entropy(A) | transform(B) | filter+colour(A, previous step)" if three else
                             "2 frame sets: entropy(set A) overlaps transform+filter(set B)") if nsets == 2 else
                ("1 frame set: entropy and transform back to back, filter+colour of step k on a second stream under the entropy launch of step k+1" if chain else "none"),
                "xyb_planes": "shared by the two sets" if nsets == 2 and not args.no_share_planes and not three and not free_running else "per frame", "parallelism": ("frames sharded over %d GPU(s), no data-path collective" % world) if args.shard == "frames" else
                (("every frame split into %d bands of group rows, one per GPU; the rows a band's filters read beyond it are exchanged after the transform stage (RCCL send / recv)" % world) if args.halo else
                 ("every frame split into %d bands of group rows, one per GPU; each GPU also decodes the group row above and below its band, no exchange" % world))},
            "roofline": {"bound": "hbm", "kernel": names[dom], "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_source,
                         "frames_per_launch": frames_per_launch, "launch_ms": round(stage_ms[dom] * frames_per_launch, 4),
                         "launch_ms_alone": round(ent_alone, 4) if dom == 0 else None,
                         # > 1: consecutive steps' launches of this kernel (one per frame set, each on its own stream) overlap
                         "launch_overlap_factor": round(stage_ms[dom] * frames_per_launch / (max_elapsed / args.steps * 1e3), 3) if dom == 0 else None,
                         "frac_alone": round(alg[names[0]] * args.batch / (ent_alone * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if dom == 0 else None,
                         "algorithmic_bytes_per_launch": int(alg[names[dom]] * frames_per_launch),
                         "note": ("entropy decode is serial per 256x256 group (latency-bound, not HBM-bound); amortised over the frames of one "
                                  "launch. launch_ms is the live average over the timed region, where consecutive steps' entropy launches "
                                  "(one per frame set, each on its own stream) overlap each other and the other two stages, so it exceeds "
                                  "ms_per_step; launch_ms_alone / frac_alone: the same launch with nothing beside it") if dom == 0 else ""},
            # the whole path against HBM: B_alg = 39.4 + bpp / 8 bytes per pixel for the three-pass formulation
            "path_roofline": {"algorithmic_bytes_per_px": round(39.4 + bpp / 8.0, 3),
                              "achieved_gbs_per_gpu": round((39.4 + bpp / 8.0) * mps * 1e6 / world / 1e9, 1),
                              "frac_of_hbm_peak": round((39.4 + bpp / 8.0) * mps * 1e6 / world / 1e9 / HBM_PEAK_GBS, 4)},
            "stage_ms_per_frame": {names[s]: round(stage_ms[s], 4) for s in range(3)},  # entropy: live, launch / frames; others: isolated
            "stage_gbs": {names[s]: round(alg[names[s]] / (stage_ms[s] * 1e-3) / 1e9, 2) for s in range(3)},
            "stage_hbm_frac": {names[s]: round(alg[names[s]] / (stage_ms[s] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) for s in range(3)},
        }
        # (the end-to-end and the CPU legs: one-GPU runs only, so that the ranks of a scaling run end together)
        if args.e2e_frames > 0 and args.shard == "frames" and world == 1:
            out["e2e"] = end_to_end(J, datas, args.e2e_frames, local_rank, xsize, ysize)
            out["e2e"]["single_image"] = single_image_api(J, data, xsize, ysize)
        if not args.no_cpu_baseline and world == 1:
            ref = system_libjxl_baseline(data, xsize, ysize, os.cpu_count() or 1)
            out["cpu_baseline"] = ref if ref else cpu_baseline(data, xsize, ysize)
        print(json.dumps(out))
    for f in frames:
        f.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
