#!/usr/bin/env python3
"""The reference's stage-3 (refinement) evaluation driver on pcdms_amd, file formats and flags unchanged.

Same command line, checkpoint layout and outputs as /root/reference/stage3_batchtest_refined_model.py (model loading :97-128, per-pair
inputs :136-157, sampling call :160-171, outputs :174-205): DINOv2 features of the source through ``ImageProjModel_p``, the stage-2
result ``<source>_to_<target>.png`` as ``vae_gen_t_image``, stock UNet with 8 input channels, N = 4 refined samples.  As in the
reference, ``guidance_rescale`` is passed the guidance scale (:163).  SSIM: see tools/stage2_batchtest_inpaint_model.py.
"""
from __future__ import annotations

import argparse
import importlib.util
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch
import torch.multiprocessing as mp
from PIL import Image

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import pcdms_amd as P  # noqa: E402

_spec = importlib.util.spec_from_file_location("stage2_driver", Path(__file__).resolve().parent / "stage2_batchtest_inpaint_model.py")
_s2 = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_s2)
to_tensor_normalized, ssim_gaussian, image_grid = _s2.to_tensor_normalized, _s2.ssim_gaussian, _s2.image_grid


def inference(args, rank, select_test_datas):
    from transformers import CLIPImageProcessor
    device = torch.device(f"cuda:{rank}")
    torch.cuda.set_device(device)
    generator = torch.Generator(device=device).manual_seed(args.seed_number)
    tag = "guidancescale{}_seed{}_numsteps{}/".format(args.guidance_scale, args.seed_number, args.num_inference_steps)
    save_dir, save_dir_metric = f"{args.save_path}/show_{tag}", f"{args.save_path}/{tag}"
    os.makedirs(save_dir, exist_ok=True)
    os.makedirs(save_dir_metric, exist_ok=True)
    clip_image_processor = CLIPImageProcessor()
    image_encoder_p = P.Dinov2Model.from_pretrained(args.image_encoder_p_path).to(device).eval()
    model_sd = torch.load("{}/mp_rank_00_model_states.pt".format(args.weights_name), map_location="cpu")["module"]
    image_proj_dict, unet_dict = {}, {}
    for k, v in model_sd.items():
        if k.startswith("image_proj_model_p"):
            image_proj_dict[k.replace("image_proj_model_p.", "")] = v
        elif k.startswith("unet"):
            unet_dict[k.replace("unet.", "")] = v
        else:
            print(k)
    hid, in_dim = image_proj_dict["net.0.weight"].shape
    image_proj_model_p = P.ImageProjModel_p(in_dim=in_dim, hidden_dim=hid, out_dim=image_proj_dict["net.4.weight"].shape[0]).to(device).eval()
    image_proj_model_p.load_state_dict(image_proj_dict)
    pipe = P.Stage3_RefinedDiffusionPipeline.from_pretrained(args.pretrained_model_name_or_path, torch_dtype=torch.float16).to(device)
    pipe.unet = P.UNet2DConditionModel.from_pretrained(args.pretrained_model_name_or_path, subfolder="unet", in_channels=8,
                                                       low_cpu_mem_usage=False, ignore_mismatched_sizes=True).to(device)
    pipe.unet.load_state_dict(unet_dict)
    pipe.scheduler = P.UniPCMultistepScheduler.from_config(pipe.scheduler.config)
    pipe.enable_xformers_memory_efficient_attention()
    print("====================== json_data: {}, model load finish ===================".format(args.json_path.split("/")[-1]))
    W, H = args.img_width, args.img_height
    all_ssim = []
    start_time = time.time()
    for data in select_test_datas:
        s_img_path = args.img_path + data["source_image"].replace(".jpg", ".png")
        t_img_path = args.img_path + data["target_image"].replace(".jpg", ".png")
        gen_t_img_path = args.gen_t_img_path + s_img_path.split("/")[-1].replace(".png", "_to_") + t_img_path.split("/")[-1]
        load = lambda p: Image.open(p).convert("RGB").resize((W, H), Image.BICUBIC)  # noqa: E731
        s_img, t_img, gen_t_img = load(s_img_path), load(t_img_path), load(gen_t_img_path)
        pix = clip_image_processor(images=s_img, return_tensors="pt").pixel_values
        s_img_proj_f = image_proj_model_p(image_encoder_p(pix.to(device)).last_hidden_state)
        vae_gen_t_image = to_tensor_normalized(gen_t_img).unsqueeze(0)
        output = pipe(height=H, width=W, guidance_rescale=args.guidance_scale, vae_gen_t_image=vae_gen_t_image, s_img_proj_f=s_img_proj_f,
                      num_images_per_prompt=4, guidance_scale=args.guidance_scale, generator=generator,
                      num_inference_steps=args.num_inference_steps)
        ssim_values = [ssim_gaussian(np.array(t_img), np.array(g)) for g in output.images]
        out_name = s_img_path.split("/")[-1].replace(".png", "") + "_to_" + t_img_path.split("/")[-1]
        if args.calculate_metrics:
            best = int(np.argmax(ssim_values))
            all_ssim.append(ssim_values[best])
            output.images[best].save(save_dir_metric + out_name)
        else:
            t_pose = load(args.pose_path + data["target_image"].replace(".jpg", "_pose.jpg"))
            image_grid([t_pose, s_img, t_img] + list(output.images), 1, 7).save(save_dir + str(min(ssim_values)) + "_" + out_name)
    print(time.time() - start_time)
    if args.calculate_metrics and all_ssim:
        print(sum(all_ssim) / len(all_ssim))
    return all_ssim


def _guidance(s: str):
    """The reference declares ``type=int, default=2.0``: the untouched default formats as ``guidancescale2.0`` and an explicit
    ``--guidance_scale 2`` as ``guidancescale2`` (which is what the stage-3 driver's default ``--gen_t_img_path`` expects).  Same
    here; non-integer values (an argparse error in the reference) are accepted as floats."""
    try:
        return int(s)
    except ValueError:
        return float(s)


def build_parser():
    p = argparse.ArgumentParser(description="Stage-3 refinement evaluation driver (reference flags) on pcdms_amd.")
    p.add_argument("--pretrained_model_name_or_path", type=str, default="./stable-diffusion-2-1-base")
    p.add_argument("--image_encoder_p_path", type=str, default="./dinov2-giant")
    p.add_argument("--img_path", type=str, default="./datasets/deepfashing/train_all_png/")
    p.add_argument("--pose_path", type=str, default="./datasets/deepfashing/openpose_all_img/")
    p.add_argument("--json_path", type=str, default="./datasets/deepfashing/test_data.json")
    p.add_argument("--gen_t_img_path", type=str, default="./save_data/stage2/guidancescale2_seed42_numsteps20/")
    p.add_argument("--save_path", type=str, default="./save_data/stage3")
    p.add_argument("--guidance_scale", type=_guidance, default=2.0)   # ref: type=int, default=2.0 => "2.0" by default, "2" when given
    p.add_argument("--seed_number", type=int, default=42)
    p.add_argument("--num_inference_steps", type=int, default=20)
    p.add_argument("--img_width", type=int, default=512)
    p.add_argument("--img_height", type=int, default=512)
    p.add_argument("--calculate_metrics", action="store_true")
    p.add_argument("--weights_name", type=str, default="./Checkpoints/stage3_checkpoints/512")
    return p


if __name__ == "__main__":
    args = build_parser().parse_args()
    print(args)
    num_devices = torch.cuda.device_count()
    datas = json.load(open(args.json_path))
    mp.set_start_method("spawn")
    chunks = P.split_list_into_chunks(datas, num_devices)
    procs = [mp.Process(target=inference, args=(args, r, chunks[r])) for r in range(num_devices)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join()
