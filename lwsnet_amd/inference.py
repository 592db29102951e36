#!/usr/bin/env python3
"""Inference CLI with the I/O contract of /root/reference/inference.py (same flags, crop rule, normalisation,
per-stage colour-mapped PNG outputs and log line), running the MI355X-native model.

    python -m lwsnet_amd.inference --left_img path/left_test.png --model checkpoint.pdparams
    python -m lwsnet_amd.inference --left_img path/left_test.png --synthetic_weights      # no checkpoint needed

Differences from the reference, all on the host side: PIL instead of cv2 (absent in this image); timing is taken
after a device synchronise and excludes the first (warm-up) call; `--vis` is accepted and ignored (no display).

`--workers N` (not in the reference; directory mode only): the reference's loop (inference.py:88-137) is one pair at a time --
decode, forward, colour-map, encode, all on one thread, which is what its published "10 FPS" measures.  With N > 0 the same
per-pair work is pipelined: N host worker PROCESSES (spawned, numpy + PIL only, no GPU) PNG-decode and crop into shared-memory
slots that are registered with HIP as pinned memory; a copy stream uploads the uint8 pixels and normalises them on the device
(lws_preprocess_rgb8: bit for bit the host transform), the forwards run through lws_pool (several batch-1 forwards in flight), a
second stream casts and colour-maps the stage-4 map on the device (lws_apply_lut8) and brings 3 bytes per pixel back into the
slot, and the same workers PNG-encode them.  The files written are byte-identical to the sequential loop's
(tests/test_gpu_parity.py::test_cli_directory_pipeline_writes_identical_files); the end-to-end rate and where the time goes
are logged and returned (profiles/r06/e2e_cli.txt).
"""
import argparse
import glob
import logging
import os
import shutil
import sys
import time

import numpy as np


def build_parser():
    p = argparse.ArgumentParser(description="Model Inference")        # inference.py:17-29
    p.add_argument("--max_disparity", type=int, default=192)
    p.add_argument("--img_path", type=str, default="dataset/kitti2015/testing/")
    p.add_argument("--left_img", type=str, default="")
    p.add_argument("--model", type=str, default="results/finetune/checkpoint.pdparams")
    p.add_argument("--save_path", type=str, default="results/inference")
    p.add_argument("--maxdisplist", type=int, nargs="+", default=[24, 5, 5])
    p.add_argument("--channels_3d", type=int, default=8)
    p.add_argument("--layers_3d", type=int, default=4)
    p.add_argument("--growth_rate", type=int, nargs="+", default=[4, 1, 1])
    p.add_argument("--gpu_id", type=int, default=0)
    p.add_argument("--vis", action="store_true", default=False)
    p.add_argument("--synthetic_weights", action="store_true",
                   help="use the seeded synthetic weights instead of --model (the reference ships no checkpoint)")
    p.add_argument("--split_bf16", action="store_true",
                   help="opt-in numerics mode of this build (not in the reference): MFMA convolutions on split-bf16 operands, "
                        "float32-level accuracy, +20-25 %% speed, not bit-identical to the default (include/lwsnet_hip.h)")
    p.add_argument("--workers", type=int, default=0,
                   help="directory mode: host worker processes for decode and encode around a pipelined GPU path (0 = the "
                        "reference's sequential loop; not in the reference)")
    p.add_argument("--gpu_workers", type=int, default=3, help="with --workers: forwards kept in flight by lws_pool")
    return p


def _host_worker(task_q, done_q, slot_names, H, W):
    """Body of a host worker PROCESS of the pipelined directory mode (spawned: a fresh interpreter that imports numpy and PIL
    only and never touches the GPU).  Tasks: ("decode", i, slot, left path, right path) -> PNG decode + crop of both images into
    the slot's shared memory as uint8 RGB (inference.py:90-100; the normalisation of :102-103 runs on the GPU,
    lws_preprocess_rgb8); ("encode", i, slot, out path) -> PNG of the slot's colour-mapped stage-4 map (inference.py:136; the
    uint8 cast and the JET table of :114-115 run on the GPU, lws_apply_lut8).  Python threads do this work at most ~16-wide (the
    interpreter lock); processes scale with the host's cores."""
    from multiprocessing import shared_memory

    from lwsnet_amd import imageio as io
    n_px = H * W * 3
    shms = {}

    def views(sid):
        if sid not in shms:
            shm = shared_memory.SharedMemory(name=slot_names[sid])
            buf = np.ndarray((3 * n_px,), np.uint8, buffer=shm.buf)
            shms[sid] = (shm, buf[:n_px].reshape(H, W, 3), buf[n_px:2 * n_px].reshape(H, W, 3), buf[2 * n_px:].reshape(H, W, 3))
        return shms[sid]

    done_q.put(("ready", -1, -1, 0.0))                                  # interpreter up, numpy and PIL imported
    while True:
        task = task_q.get()
        if task is None:
            break
        kind, i, sid = task[0], task[1], task[2]
        t0 = time.perf_counter()
        try:
            if kind == "decode":
                left = io.crop_bottom_right(io.load_rgb(task[3]))
                right = io.crop_bottom_right(io.load_rgb(task[4]))
                if left is None or right is None:                       # inference.py:96-97
                    done_q.put(("skipped", i, sid, 0.0))
                    continue
                _, vl, vr, _ = views(sid)
                np.copyto(vl, left)
                np.copyto(vr, right)
                done_q.put(("decoded", i, sid, time.perf_counter() - t0))
            else:
                io.save_png(task[3], views(sid)[3])
                done_q.put(("encoded", i, sid, time.perf_counter() - t0))
        except Exception as e:                                          # noqa: BLE001 (reported to the parent, which raises)
            done_q.put(("error", i, sid, f"{kind} of pair {i}: {type(e).__name__}: {e}"))
    for shm, *_ in shms.values():
        shm.close()


class _Slot:
    """Buffers of one pair in flight: a shared-memory block [left RGB | right RGB | colour-mapped stage-4 map], all uint8 HWC,
    that the host workers write and read -- registered with HIP as pinned memory when the runtime allows (otherwise staged
    through pinned tensors) --, the device copies of the three images, the normalised device inputs and the four stage maps."""

    def __init__(self, dev, H, W):
        from multiprocessing import shared_memory

        import torch
        n_px = H * W * 3
        self.shm = shared_memory.SharedMemory(create=True, size=3 * n_px)
        host = torch.frombuffer(self.shm.buf, dtype=torch.uint8)
        self.registered = False
        try:
            if os.environ.get("LWS_CLI_NO_HOST_REGISTER") != "1":      # (tests force the staging path with it)
                rc = torch.cuda.cudart().cudaHostRegister(host.data_ptr(), host.numel(), 0)
                self.registered = int(rc) == 0 and host.is_pinned()
        except Exception:                                               # noqa: BLE001 (fall back to staging copies)
            self.registered = False
        self.host_in, self.host_out = host[:2 * n_px].view(2, H, W, 3), host[2 * n_px:].view(H, W, 3)
        if not self.registered:
            self.pin_in = torch.empty((2, H, W, 3), dtype=torch.uint8).pin_memory()
            self.pin_out = torch.empty((H, W, 3), dtype=torch.uint8).pin_memory()
        self.dev_in = torch.empty((2, H, W, 3), dtype=torch.uint8, device=dev)
        self.dev_rgb = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
        self.dev_lr = torch.empty((2, 3, H, W), dtype=torch.float32, device=dev)      # [left | right], normalised
        self.outs = [torch.empty((1, 1, H, W), dtype=torch.float32, device=dev) for _ in range(4)]
        self.ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]      # h2d begin / end, d2h begin / end
        self.t0 = 0.0

    def close(self):
        import torch
        ptr = self.host_in.data_ptr()
        self.host_in = self.host_out = None
        if self.registered:
            try:
                torch.cuda.cudart().cudaHostUnregister(ptr)
            except Exception:                                           # noqa: BLE001
                pass
        try:
            self.shm.close()
        except BufferError:                                             # a view is still alive somewhere: unlink anyway
            pass
        self.shm.unlink()


def inference_pipelined(model, left_imgs, right_imgs, args, log):
    """The loop of inference.py:88-137 in directory mode, pipelined (see the module docstring).  Returns (written, stats)."""
    import multiprocessing as mp
    import queue
    import threading

    import torch
    from . import imageio as io
    from . import ops
    dev = model.device
    N, P = max(1, int(args.workers)), max(1, int(args.gpu_workers))
    H, W = io.CROP_H, io.CROP_W
    total = len(left_imgs)
    torch.cuda.set_device(dev)
    slots = [_Slot(dev, H, W) for _ in range(2 * P + 2 * N)]
    lut_dev = torch.from_numpy(io.jet_lut()).to(dev)
    ctx = mp.get_context("spawn")                        # fresh interpreters: a forked child of a process that holds HIP state is not safe
    task_q, done_q = ctx.Queue(), ctx.Queue()
    names = [sl.shm.name for sl in slots]
    procs = [ctx.Process(target=_host_worker, args=(task_q, done_q, names, H, W), daemon=True) for _ in range(N)]
    for pr in procs:
        pr.start()
    free = queue.Queue()
    for sid in range(len(slots)):
        free.put(sid)
    inflight = queue.Queue()
    written = {}
    acc = {"decode_s": 0.0, "encode_s": 0.0, "h2d_ms": 0.0, "d2h_ms": 0.0, "pairs": 0, "skipped": 0, "latency_s": 0.0}
    errors = []
    h2d, d2h = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)

    def feeder():
        for i in range(total):
            sid = free.get()
            slots[sid].t0 = time.perf_counter()
            task_q.put(("decode", i, sid, left_imgs[i], right_imgs[i]))

    def collector():
        torch.cuda.set_device(dev)
        while True:
            item = inflight.get()
            if item is None:
                return
            i, sid, job = item
            sl = slots[sid]
            try:
                job.result()                                            # the four stage maps are complete in device memory
                dst = sl.host_out if sl.registered else sl.pin_out
                with torch.cuda.stream(d2h):
                    sl.ev[2].record()
                    ops.apply_lut8(sl.outs[3][0, 0], lut_dev, out=sl.dev_rgb)   # inference.py:114-115; directory mode keeps stage 4 (:133-137)
                    dst.copy_(sl.dev_rgb, non_blocking=True)
                    sl.ev[3].record()
                sl.ev[3].synchronize()
                if not sl.registered:
                    sl.host_out.copy_(sl.pin_out)
                task_q.put(("encode", i, sid, os.path.join(args.save_path, os.path.basename(left_imgs[i]))))
            except Exception as e:                                      # noqa: BLE001
                errors.append(e)
                done_q.put(("error", i, sid, repr(e)))

    wall = 0.0
    try:
        # warm-up outside the clock (the reference times its first call; this build never does): library, pool and workers up
        with model.pool(workers=P) as gpool:
            gpool.reserve(1, H, W)
            gpool.submit(slots[0].dev_lr[:1].zero_(), slots[0].dev_lr[1:].zero_(), out=slots[0].outs).result()
            torch.cuda.synchronize(dev)
            ready, t_wait = 0, time.perf_counter()
            while ready < N:                                            # every worker has started (spawn + imports: ~1 s, once)
                try:
                    msg = done_q.get(timeout=5.0)
                except queue.Empty:
                    if not all(pr.is_alive() for pr in procs) or time.perf_counter() - t_wait > 120.0:
                        raise RuntimeError("the host worker processes did not start")
                    continue
                if msg[0] == "ready":
                    ready += 1
                else:
                    raise RuntimeError(f"unexpected message from a host worker before its start-up: {msg[:3]}")
            t_begin = time.perf_counter()
            tf = threading.Thread(target=feeder, daemon=True)
            tc = threading.Thread(target=collector, daemon=True)
            tf.start()
            tc.start()
            done = 0
            while done < total and not errors:
                try:
                    kind, i, sid, val = done_q.get(timeout=5.0)
                except queue.Empty:
                    if not all(pr.is_alive() for pr in procs):
                        errors.append(RuntimeError("a host worker process died"))
                    continue
                sl = slots[sid]
                if kind == "decoded":
                    acc["decode_s"] += val
                    src = sl.host_in if sl.registered else sl.pin_in
                    if not sl.registered:
                        sl.pin_in.copy_(sl.host_in)
                    with torch.cuda.stream(h2d):
                        sl.ev[0].record()
                        sl.dev_in.copy_(src, non_blocking=True)
                        ops.preprocess_rgb8(sl.dev_in, out=sl.dev_lr)           # inference.py:102-103 (ToTensor + Normalize)
                        sl.ev[1].record()
                        job = gpool.submit(sl.dev_lr[:1], sl.dev_lr[1:], out=sl.outs)   # starts behind them (after_stream = h2d)
                    inflight.put((i, sid, job))
                elif kind == "encoded":
                    acc["encode_s"] += val
                    acc["h2d_ms"] += sl.ev[0].elapsed_time(sl.ev[1])
                    acc["d2h_ms"] += sl.ev[2].elapsed_time(sl.ev[3])
                    acc["pairs"] += 1
                    acc["latency_s"] += time.perf_counter() - sl.t0
                    written[i] = os.path.join(args.save_path, os.path.basename(left_imgs[i]))
                    free.put(sid)
                    done += 1
                elif kind == "skipped":
                    acc["skipped"] += 1
                    free.put(sid)
                    done += 1
                else:
                    errors.append(RuntimeError(val))
            wall = time.perf_counter() - t_begin
            inflight.put(None)
            tc.join(timeout=30.0)
    finally:
        for _ in procs:
            task_q.put(None)
        for pr in procs:
            pr.join(timeout=10.0)
            if pr.is_alive():
                pr.terminate()                                          # (the exact children started above)
        registered = all(sl.registered for sl in slots)
        torch.cuda.synchronize(dev)
        for sl in slots:
            sl.close()
    if errors:
        raise errors[0]
    n = max(acc["pairs"], 1)
    stats = {"pairs": acc["pairs"], "skipped": acc["skipped"], "wall_s": round(wall, 4), "pairs_per_s": round(acc["pairs"] / wall, 2),
             "host_processes": N, "gpu_workers": P, "shared_memory_pinned": registered,
             "decode_ms_per_pair": round(1e3 * acc["decode_s"] / n, 3), "encode_ms_per_pair": round(1e3 * acc["encode_s"] / n, 3),
             "h2d_ms_per_pair": round(acc["h2d_ms"] / n, 3), "d2h_ms_per_pair": round(acc["d2h_ms"] / n, 3),
             "latency_ms_per_pair": round(1e3 * acc["latency_s"] / n, 2)}
    paths = [written[i] for i in sorted(written)]
    for pth in paths:
        log.info("Inference 4 stages cost = {:.3f} sec, FPS = {:.1f}\t\tSave img = {}".format(wall / n, n / wall, pth))
    log.info("pipelined: %s", stats)
    return paths, stats


def inference(model, left_imgs, right_imgs, args, log):
    """inference.py:78-138."""
    import torch
    from . import imageio as io
    written = []
    warm = False
    for li, ri in zip(left_imgs, right_imgs):
        left = io.crop_bottom_right(io.load_rgb(li))
        right = io.crop_bottom_right(io.load_rgb(ri))
        if left is None or right is None:                               # :96-97
            continue
        l_in, r_in = io.to_input(left)[None], io.to_input(right)[None]
        if not warm:                                                    # one warm-up in all (the reference times its first call)
            model(l_in, r_in)
            warm = True
        torch.cuda.synchronize(model.device)
        t0 = time.time()
        outputs = model(l_in, r_in)
        torch.cuda.synchronize(model.device)
        cost = time.time() - t0
        ss = "Inference 4 stages cost = {:.3f} sec, FPS = {:.1f}".format(cost, 1 / cost)
        color = None
        for stage in range(4):
            disp = outputs[stage].squeeze(axis=[0, 1]).numpy()          # :114 (the uint8 cast is inside disparity_to_color)
            color = io.disparity_to_color(disp)
            if args.left_img:                                           # :117-122
                path = os.path.join(os.path.dirname(args.left_img), str(stage + 1) + ".png")
                io.save_png(path, color)
                written.append(path)
                log.info("{}\t\tSave img = {}".format(ss, path))
        if not args.left_img:                                           # :133-137 (stage-4 map only)
            path = os.path.join(args.save_path, os.path.basename(li))
            io.save_png(path, color)
            written.append(path)
            log.info("{}\t\tSave img = {}".format(ss, path))
    return written


def main(argv=None):
    args = build_parser().parse_args(argv)
    logging.basicConfig(stream=sys.stderr, level=logging.INFO,
                        format="[%(asctime)s %(filename)s:%(lineno)s] %(levelname)s: %(message)s")
    log = logging.getLogger("lwsnet_amd.inference")
    for k, v in vars(args).items():
        log.info("%s: %s", k, v)
    import torch
    from .checkpoint import load_state_dict
    from .models import LWSNet
    from .weights import make_state_dict
    torch.cuda.set_device(args.gpu_id)                                  # inference.py:38
    model = LWSNet(args, device=torch.device("cuda", args.gpu_id))
    if args.synthetic_weights:
        model.set_state_dict(make_state_dict(7, args))
        log.info("Using seeded synthetic weights")
    elif not os.path.isfile(args.model):                                # inference.py:41-43
        log.info("No model load")
        raise SystemExit
    else:
        model.set_state_dict(load_state_dict(args.model))
        log.info("Successful load model")
    model.eval()
    if getattr(args, "split_bf16", False):
        model.set_option("split_bf16", 7)
        log.info("split-bf16 numerics mode")
    if not args.left_img:                                               # :50-63
        if os.path.isdir(args.img_path):
            lefts = sorted(glob.glob(os.path.join(args.img_path, "image_2/*.png")))
            rights = sorted(glob.glob(os.path.join(args.img_path, "image_3/*.png")))
        else:
            base, name = os.path.dirname(os.path.dirname(args.img_path)), os.path.basename(args.img_path)
            lefts, rights = [os.path.join(base, "image_2", name)], [os.path.join(base, "image_3", name)]
        if os.path.exists(args.save_path):
            shutil.rmtree(args.save_path)
        os.makedirs(args.save_path)
    else:                                                               # :65-70
        lefts = [args.left_img]
        rights = [os.path.join(os.path.dirname(args.left_img), "right_test.png")]
    log.info("Begin inference!")
    if args.workers > 0 and not args.left_img:
        written, stats = inference_pipelined(model, lefts, rights, args, log)
        main.last_stats = stats
    else:
        written = inference(model, lefts, rights, args, log)
    log.info("End inference!")
    return written


if __name__ == "__main__":
    main()
