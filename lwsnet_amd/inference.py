#!/usr/bin/env python3
"""Inference CLI with the I/O contract of /root/reference/inference.py (same flags, crop rule, normalisation,
per-stage colour-mapped PNG outputs and log line), running the MI355X-native model.

    python -m lwsnet_amd.inference --left_img path/left_test.png --model checkpoint.pdparams
    python -m lwsnet_amd.inference --left_img path/left_test.png --synthetic_weights      # no checkpoint needed

Differences from the reference, all on the host side: PIL instead of cv2 (absent in this image); timing is taken
after a device synchronise and excludes the first (warm-up) call; `--vis` is accepted and ignored (no display).
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
    return p


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
    written = inference(model, lefts, rights, args, log)
    log.info("End inference!")
    return written


if __name__ == "__main__":
    main()
