#!/usr/bin/env python3
"""Inference CLI with the I/O contract of /root/reference/inference.py (same flags, crop rule, normalisation,
per-stage colour-mapped PNG outputs and log line), running the MI355X-native model.

    python -m lwsnet_amd.inference --left_img path/left_test.png --model checkpoint.pdparams
    python -m lwsnet_amd.inference --left_img path/left_test.png --synthetic_weights      # no checkpoint needed

Differences from the reference, all on the host side: PIL instead of cv2 (absent in this image); timing is taken
after a device synchronise and excludes the first (warm-up) call; `--vis` is accepted and ignored (no display).

`--workers N` (not in the reference; directory mode only): the reference's loop (inference.py:88-137) is one pair at a time --
decode, forward, colour-map, encode, all on one thread, which is what its published "10 FPS" measures.  With N > 0 the same
per-pair work is pipelined: N host threads decode / crop / normalise into pinned buffers, a copy stream uploads them, the
forwards run through lws_pool (several batch-1 forwards in flight), a second copy stream brings the stage-4 maps back and N
host threads colour-map and PNG-encode them.  The files written are byte-identical to the sequential loop's
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
                   help="directory mode: host threads for decode and for encode around a pipelined GPU path (0 = the reference's "
                        "sequential loop; not in the reference)")
    p.add_argument("--gpu_workers", type=int, default=3, help="with --workers: forwards kept in flight by lws_pool")
    return p


class _Slot:
    """Buffers of one pair in flight: pinned host inputs, device inputs, device stage maps, pinned stage-4 map."""

    def __init__(self, dev, H, W):
        import torch
        self.pin_l = torch.empty((1, 3, H, W), dtype=torch.float32).pin_memory()
        self.pin_r = torch.empty((1, 3, H, W), dtype=torch.float32).pin_memory()
        self.dev_l = torch.empty((1, 3, H, W), dtype=torch.float32, device=dev)
        self.dev_r = torch.empty((1, 3, H, W), dtype=torch.float32, device=dev)
        self.outs = [torch.empty((1, 1, H, W), dtype=torch.float32, device=dev) for _ in range(4)]
        self.pin_out = torch.empty((H, W), dtype=torch.float32).pin_memory()
        self.ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]      # h2d begin / end, d2h begin / end


def inference_pipelined(model, left_imgs, right_imgs, args, log):
    """The loop of inference.py:88-137 in directory mode, pipelined (see the module docstring).  Returns (written, stats)."""
    import concurrent.futures as cf
    import queue
    import threading

    import torch
    from . import imageio as io
    dev = model.device
    N, P = max(1, int(args.workers)), max(1, int(args.gpu_workers))
    H, W = io.CROP_H, io.CROP_W
    nslots = 2 * P + N
    torch.cuda.set_device(dev)
    slots = [_Slot(dev, H, W) for _ in range(nslots)]
    free = queue.Queue()
    for sl in slots:
        free.put(sl)
    ready, inflight = queue.Queue(), queue.Queue()
    written, lock = {}, threading.Lock()
    acc = {"decode_s": 0.0, "encode_s": 0.0, "h2d_ms": 0.0, "d2h_ms": 0.0, "pairs": 0, "skipped": 0, "latency_s": 0.0}
    errors = []
    h2d, d2h = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)

    def decode(i, sl):
        try:
            t0 = time.perf_counter()
            left = io.crop_bottom_right(io.load_rgb(left_imgs[i]))
            right = io.crop_bottom_right(io.load_rgb(right_imgs[i]))
            if left is None or right is None:                           # inference.py:96-97
                with lock:
                    acc["skipped"] += 1
                free.put(sl)
                ready.put(None)
                return
            np.copyto(sl.pin_l.numpy()[0], io.to_input(left))
            np.copyto(sl.pin_r.numpy()[0], io.to_input(right))
            with lock:
                acc["decode_s"] += time.perf_counter() - t0
            ready.put((i, sl, t0))
        except Exception as e:                                          # noqa: BLE001 (re-raised by the caller)
            errors.append(e)
            ready.put(None)

    def encode(i, sl, t_start):
        try:
            t0 = time.perf_counter()
            color = io.disparity_to_color(sl.pin_out.numpy())
            path = os.path.join(args.save_path, os.path.basename(left_imgs[i]))
            io.save_png(path, color)
            t1 = time.perf_counter()
            with lock:
                acc["encode_s"] += t1 - t0
                acc["h2d_ms"] += sl.ev[0].elapsed_time(sl.ev[1])
                acc["d2h_ms"] += sl.ev[2].elapsed_time(sl.ev[3])
                acc["pairs"] += 1
                acc["latency_s"] += t1 - t_start
                written[i] = path
            free.put(sl)
        except Exception as e:                                          # noqa: BLE001
            errors.append(e)
            free.put(sl)

    def feeder(pool_):
        for i in range(len(left_imgs)):
            sl = free.get()
            pool_.submit(decode, i, sl)

    def collector(enc_pool):
        torch.cuda.set_device(dev)
        while True:
            item = inflight.get()
            if item is None:
                return
            i, sl, job, t0 = item
            try:
                job.result()                                            # the four stage maps are complete in device memory
                with torch.cuda.stream(d2h):
                    sl.ev[2].record()
                    sl.pin_out.copy_(sl.outs[3][0, 0], non_blocking=True)   # directory mode keeps the stage-4 map only (:133-137)
                    sl.ev[3].record()
                sl.ev[3].synchronize()
                enc_pool.submit(encode, i, sl, t0)
            except Exception as e:                                      # noqa: BLE001
                errors.append(e)
                free.put(sl)

    # warm-up outside the clock (the reference times its first call; this build never does): one forward, library and pool up
    with model.pool(workers=P) as gpool:
        gpool.reserve(1, H, W)
        gpool.submit(slots[0].dev_l.zero_(), slots[0].dev_r.zero_(), out=slots[0].outs).result()
        torch.cuda.synchronize(dev)
        t_begin = time.perf_counter()
        with cf.ThreadPoolExecutor(N, thread_name_prefix="lws-decode") as dec_pool, \
                cf.ThreadPoolExecutor(N, thread_name_prefix="lws-encode") as enc_pool:
            tf = threading.Thread(target=feeder, args=(dec_pool,), daemon=True)
            tc = threading.Thread(target=collector, args=(enc_pool,), daemon=True)
            tf.start()
            tc.start()
            for _ in range(len(left_imgs)):
                item = ready.get()
                if item is None:
                    continue
                i, sl, t0 = item
                with torch.cuda.stream(h2d):
                    sl.ev[0].record()
                    sl.dev_l.copy_(sl.pin_l, non_blocking=True)
                    sl.dev_r.copy_(sl.pin_r, non_blocking=True)
                    sl.ev[1].record()
                    job = gpool.submit(sl.dev_l, sl.dev_r, out=sl.outs)     # starts behind the copies (after_stream = h2d)
                inflight.put((i, sl, job, t0))
            inflight.put(None)
            tf.join()
            tc.join()
        wall = time.perf_counter() - t_begin                                # (the executors' exit waits for the last encode)
    if errors:
        raise errors[0]
    n = max(acc["pairs"], 1)
    stats = {"pairs": acc["pairs"], "skipped": acc["skipped"], "wall_s": round(wall, 4), "pairs_per_s": round(acc["pairs"] / wall, 2),
             "host_threads": N, "gpu_workers": P,
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
