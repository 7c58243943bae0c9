#!/usr/bin/env python3
"""The reference's stage-1 (prior) evaluation driver on pcdms_amd, file formats and flags unchanged.

Same command line, checkpoint layout and outputs as /root/reference/stage1_batchtest_prior_model.py (flags :140-155, model loading
:52-62, per-pair inputs :74-98, sampling call :105-113, feature .npy + cosine similarity :115-135, one process per GPU :171-182): CLIP
ViT-H/14 image embedding of the source, the two 18-keypoint pose files, the UnCLIP-sampled prior, ``<source>_to_<target>.npy`` out
(what the stage-2 driver's ``--target_embed_path`` reads).  Every model object is the pcdms_amd one.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch
import torch.multiprocessing as mp
import torch.nn.functional as F
from PIL import Image

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import pcdms_amd as P  # noqa: E402


def read_coordinates_file(file_path) -> torch.Tensor:
    """one "x y" pair per line -> [1, 2 * n_keypoints] (:20-27)."""
    vals = []
    with open(file_path) as f:
        for line in f:
            x, y = map(float, line.strip().split())
            vals.extend([x, y])
    return torch.tensor(vals, dtype=torch.float32).view(1, -1)


def main(args, rank, select_test_datas):
    from transformers import CLIPImageProcessor
    device = torch.device(f"cuda:{rank}")
    torch.cuda.set_device(device)
    generator = torch.Generator(device=device).manual_seed(args.seed_number)
    save_dir = "{}/guidancescale{}_seed{}_numsteps{}/".format(args.save_path, args.guidance_scale, args.seed_number, args.num_inference_steps)
    os.makedirs(save_dir, exist_ok=True)
    clip_image_processor = CLIPImageProcessor()
    pipe = P.Stage1_PriorPipeline.from_pretrained(args.pretrained_model_name_or_path).to(device)
    pipe.prior = P.Stage1_PriorTransformer.from_pretrained(args.pretrained_model_name_or_path, subfolder="prior", num_embeddings=2,
                                                           embedding_dim=1024, low_cpu_mem_usage=False, ignore_mismatched_sizes=True).to(device)
    prior_dict = torch.load("{}/mp_rank_00_model_states.pt".format(args.weights_name), map_location="cpu")["module"]
    pipe.prior.load_state_dict(prior_dict)
    pipe.enable_xformers_memory_efficient_attention()
    image_encoder = P.CLIPVisionModelWithProjection.from_pretrained(args.image_encoder_path).eval().to(device)
    print("====================== model load finish ===================")
    start_time = time.time()
    sims = []
    for d in select_test_datas:
        s_img_path = args.img_path + d["source_image"].replace(".jpg", ".png")
        t_img_path = args.img_path + d["target_image"].replace(".jpg", ".png")
        s_pose = read_coordinates_file(args.pose_path + d["source_image"].replace(".jpg", ".txt")).to(device).unsqueeze(1)
        t_pose = read_coordinates_file(args.pose_path + d["target_image"].replace(".jpg", ".txt")).to(device).unsqueeze(1)
        load = lambda p: Image.open(p).convert("RGB").resize((args.img_width, args.img_height), Image.BICUBIC)  # noqa: E731
        clip_s = clip_image_processor(images=load(s_img_path), return_tensors="pt").pixel_values
        clip_t = clip_image_processor(images=load(t_img_path), return_tensors="pt").pixel_values
        s_img_embed = image_encoder(clip_s.to(device)).image_embeds.unsqueeze(1)
        target_embed = image_encoder(clip_t.to(device)).image_embeds
        output = pipe(s_embed=s_img_embed, s_pose=s_pose, t_pose=t_pose, num_images_per_prompt=1, num_inference_steps=args.num_inference_steps,
                      generator=generator, guidance_scale=args.guidance_scale)
        name = s_img_path.split("/")[-1].replace(".png", "_to_") + t_img_path.split("/")[-1].replace(".png", ".npy")
        np.save(save_dir + name, output[0].cpu().detach().numpy())
        sims.append(F.cosine_similarity(output[0].float(), target_embed.float()).item())
    print(time.time() - start_time)
    avg = sum(sims) / max(len(sims), 1)
    with open(save_dir + "/a_results.txt", "a") as ff:
        ff.write("number is {}, guidance_scale is {}, all averge simm is :{} \n".format(len(sims), args.guidance_scale, avg))
    print("number is {}, guidance_scale is {}, all averge simm is :{}".format(len(sims), args.guidance_scale, avg))
    return sims


def build_parser():
    p = argparse.ArgumentParser(description="Stage-1 prior evaluation driver (reference flags) on pcdms_amd.")
    p.add_argument("--pretrained_model_name_or_path", type=str, default="./kandinsky-2-2-prior")
    p.add_argument("--image_encoder_path", type=str, default="./OpenCLIP-ViT-H-14")
    p.add_argument("--img_path", type=str, default="./datasets/deepfashing/train_all_png/")
    p.add_argument("--pose_path", type=str, default="./datasets/deepfashing/normalized_pose_txt/")
    p.add_argument("--json_path", type=str, default="./datasets/deepfashing/test_data.json")
    p.add_argument("--save_path", type=str, default="./save_data/stage1")
    p.add_argument("--guidance_scale", type=int, default=0)
    p.add_argument("--seed_number", type=int, default=42)
    p.add_argument("--num_inference_steps", type=int, default=20)
    p.add_argument("--img_width", type=int, default=512)
    p.add_argument("--img_height", type=int, default=512)
    p.add_argument("--weights_name", type=str, default="./Checkpoints/stage1_checkpoints/512")
    return p


if __name__ == "__main__":
    args = build_parser().parse_args()
    print(args)
    num_devices = torch.cuda.device_count()
    print("Using {} GPUs inference".format(num_devices))
    datas = json.load(open(args.json_path))
    print("The number of test data: {}".format(len(datas)))
    mp.set_start_method("spawn")
    chunks = P.split_list_into_chunks(datas, num_devices)
    procs = [mp.Process(target=main, args=(args, r, chunks[r])) for r in range(num_devices)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join()
