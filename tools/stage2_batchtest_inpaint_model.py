#!/usr/bin/env python3
"""The reference's stage-2 evaluation driver on pcdms_amd, file formats and flags unchanged.

Same command line, checkpoint layout and outputs as /root/reference/stage2_batchtest_inpaint_model.py (flags :242-262, model
loading :95-133, per-pair conditioning :141-186, sampling call :188-200, best-SSIM / grid output :203-232, one process per GPU
over ``split_list_into_chunks`` :266-285); every model object is the pcdms_amd one, so the whole driver -- encoders, pose
embedding, VAE, UNet, scheduler -- runs on the MI355X through libpcdm.so.  What stays host-side is what the reference does
on the host too: PIL resize / canvas pasting, ``CLIPImageProcessor``, PNG writing, SSIM.

Differences, on purpose: ``--img_width`` is honoured (the reference parses it but reads ``args.img_weigh``); ``ImageProjModel_p``
takes its sizes from the checkpoint / encoder config instead of the literals 1536 / 768 / 1024 (identical for the published
checkpoints); SSIM is a numpy restatement of ``skimage.metrics.structural_similarity`` with the driver's arguments (skimage is not
in the image); per-rank work is the reference's block partition.
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
from PIL import Image

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import pcdms_amd as P  # noqa: E402


def to_tensor_normalized(img: Image.Image) -> torch.Tensor:
    """transforms.Compose([ToTensor(), Normalize([0.5], [0.5])]) (:86-89)."""
    x = torch.from_numpy(np.asarray(img, dtype=np.float32) / 255.0).permute(2, 0, 1)
    return (x - 0.5) / 0.5


def ssim_gaussian(a: np.ndarray, b: np.ndarray, sigma: float = 1.2) -> float:
    """skimage.metrics.structural_similarity(a, b, gaussian_weights=True, sigma=1.2, use_sample_covariance=False, channel_axis=2,
    data_range=b.max() - b.min()) as the driver calls it (:206-211), restated."""
    from scipy.ndimage import gaussian_filter
    a, b = a.astype(np.float64), b.astype(np.float64)
    R = b.max() - b.min()
    c1, c2 = (0.01 * R) ** 2, (0.03 * R) ** 2
    pad = (2 * int(3.5 * sigma + 0.5) + 1 - 1) // 2
    vals = []
    for ch in range(a.shape[2]):
        x, y = a[..., ch], b[..., ch]
        f = lambda z: gaussian_filter(z, sigma, truncate=3.5, mode="reflect")  # noqa: E731
        ux, uy = f(x), f(y)
        vx, vy, vxy = f(x * x) - ux * ux, f(y * y) - uy * uy, f(x * y) - ux * uy
        s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux ** 2 + uy ** 2 + c1) * (vx + vy + c2))
        vals.append(s[pad:-pad, pad:-pad].mean())
    return float(np.mean(vals))


def image_grid(imgs, rows, cols):
    w, h = imgs[0].size
    grid = Image.new("RGB", size=(cols * w, rows * h))
    for i, img in enumerate(imgs):
        grid.paste(img, box=(i % cols * w, i // cols * h))
    return grid


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

    # ---- models (:95-133)
    image_encoder_g = P.CLIPVisionModelWithProjection.from_pretrained(args.image_encoder_g_path).to(device).eval()
    image_encoder_p = P.Dinov2Model.from_pretrained(args.image_encoder_p_path).to(device).eval()
    model_sd = torch.load("{}/mp_rank_00_model_states.pt".format(args.weights_name), map_location="cpu")["module"]
    pose_proj_dict, image_proj_dict, unet_dict = {}, {}, {}
    for k, v in model_sd.items():
        if k.startswith("pose_proj"):
            pose_proj_dict[k.replace("pose_proj.", "")] = v
        elif k.startswith("image_proj_model_p"):
            image_proj_dict[k.replace("image_proj_model_p.", "")] = v
        elif k.startswith("unet"):
            unet_dict[k.replace("unet.", "")] = v
        else:
            print(k)
    hid, in_dim = image_proj_dict["net.0.weight"].shape
    image_proj_model_p = P.ImageProjModel_p(in_dim=in_dim, hidden_dim=hid, out_dim=image_proj_dict["net.4.weight"].shape[0]).to(device).eval()
    pose_proj = P.ControlNetConditioningEmbedding(pose_proj_dict["conv_out.weight"].shape[0], 3, (16, 32, 96, 256)).to(device).eval()
    pose_proj.load_state_dict(pose_proj_dict)
    image_proj_model_p.load_state_dict(image_proj_dict)

    pipe = P.Stage2_InpaintDiffusionPipeline.from_pretrained(args.pretrained_model_name_or_path, torch_dtype=torch.float16).to(device)
    pipe.unet = P.Stage2_InapintUNet2DConditionModel.from_pretrained(
        args.pretrained_model_name_or_path, subfolder="unet", in_channels=9, class_embed_type="projection",
        projection_class_embeddings_input_dim=unet_dict["class_embedding.linear_1.weight"].shape[1], torch_dtype=torch.float16,
        low_cpu_mem_usage=False, ignore_mismatched_sizes=True).to(device)
    pipe.unet.load_state_dict(unet_dict)
    pipe.scheduler = P.UniPCMultistepScheduler.from_config(pipe.scheduler.config)
    pipe.enable_xformers_memory_efficient_attention()
    print("====================== json_data: {}, model load finish ===================".format(args.json_path.split("/")[-1]))

    W, H = args.img_width, args.img_height
    all_ssim = []
    start_time = time.time()
    split = args.json_path.split("/")[-1].split("_")[0]
    for data in select_test_datas:
        s_img_path = args.img_path + data["source_image"].replace(".jpg", ".png")
        t_img_path = args.img_path + data["target_image"].replace(".jpg", ".png")
        s_pose_path = args.pose_path + data["source_image"].replace(".jpg", "_pose.jpg")
        t_pose_path = args.pose_path + data["target_image"].replace(".jpg", "_pose.jpg")
        load = lambda p: Image.open(p).convert("RGB").resize((W, H), Image.BICUBIC)  # noqa: E731
        s_img, t_img, s_pose, t_pose = load(s_img_path), load(t_img_path), load(s_pose_path), load(t_pose_path)
        s_img_t_mask = Image.new("RGB", (2 * W, H))          # [source | black]
        s_img_t_mask.paste(s_img, (0, 0))
        st_pose = Image.new("RGB", (2 * W, H))               # [source pose | target pose]
        st_pose.paste(s_pose, (0, 0))
        st_pose.paste(t_pose, (W, 0))

        pix = clip_image_processor(images=s_img, return_tensors="pt").pixel_values
        s_img_proj_f = image_proj_model_p(image_encoder_p(pix.to(device)).last_hidden_state)
        vae_image = to_tensor_normalized(s_img_t_mask).unsqueeze(0)
        st_pose_f = pose_proj(to_tensor_normalized(st_pose).unsqueeze(0).to(device))
        if split == "train":
            pix_t = clip_image_processor(images=t_img, return_tensors="pt").pixel_values
            pred_t_img_embed = image_encoder_g(pix_t.to(device)).image_embeds.unsqueeze(1)
        elif split == "test":
            name = s_img_path.split("/")[-1].replace(".png", "_to_") + t_img_path.split("/")[-1].replace(".png", ".npy")
            pred_t_img_embed = torch.tensor(np.load(args.target_embed_path + name)).to(device).unsqueeze(1)
        else:
            raise ValueError("Check the input JSON file path")

        output = pipe(height=H, width=2 * W, guidance_rescale=0.0, vae_image=vae_image, s_img_proj_f=s_img_proj_f, st_pose_f=st_pose_f,
                      pred_t_img_embed=pred_t_img_embed, num_images_per_prompt=4, guidance_scale=args.guidance_scale, generator=generator,
                      num_inference_steps=args.num_inference_steps)
        out_name = s_img_path.split("/")[-1].replace(".png", "") + "_to_" + t_img_path.split("/")[-1]
        if args.calculate_metrics:
            ssim_values = []
            for gen_img in output.images:
                gen = np.array(gen_img.crop((W, 0, 2 * W, H))) * 255.0
                ssim_values.append(ssim_gaussian(np.array(t_img) * 255.0, gen))
            best = int(np.argmax(ssim_values))
            all_ssim.append(ssim_values[best])
            output.images[best].crop((W, 0, 2 * W, H)).save(save_dir_metric + out_name)
        else:
            vis_pose = Image.new("RGB", (2 * W, H))
            vis_pose.paste(s_pose, (0, 0))
            vis_pose.paste(t_pose, (W, 0))
            vis_img = Image.new("RGB", (2 * W, H))
            vis_img.paste(s_img, (0, 0))
            vis_img.paste(t_img, (W, 0))
            image_grid([vis_img, vis_pose] + list(output.images), 2, 3).save(save_dir + out_name)
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
    p = argparse.ArgumentParser(description="Stage-2 inpaint evaluation driver (reference flags) on pcdms_amd.")
    p.add_argument("--pretrained_model_name_or_path", type=str, default="./stable-diffusion-2-1-base")
    p.add_argument("--image_encoder_g_path", type=str, default="./OpenCLIP-ViT-H-14")
    p.add_argument("--image_encoder_p_path", type=str, default="./dinov2-giant")
    p.add_argument("--img_path", type=str, default="./datasets/deepfashing/train_all_png/")
    p.add_argument("--pose_path", type=str, default="./datasets/deepfashing/openpose_all_img/")
    p.add_argument("--json_path", type=str, default="./datasets/deepfashing/test_data.json")
    p.add_argument("--target_embed_path", type=str, default="./save_data/stage1/guidancescale0_seed42_numsteps20/")
    p.add_argument("--save_path", type=str, default="./save_data/stage2")
    p.add_argument("--guidance_scale", type=_guidance, default=2.0)   # ref: type=int, default=2.0 => "2.0" by default, "2" when given
    p.add_argument("--seed_number", type=int, default=42)
    p.add_argument("--num_inference_steps", type=int, default=20)
    p.add_argument("--img_width", type=int, default=512)
    p.add_argument("--img_height", type=int, default=512)
    p.add_argument("--calculate_metrics", action="store_true")
    p.add_argument("--weights_name", type=str, default="./Checkpoints/stage2_checkpoints/512")
    return p


if __name__ == "__main__":
    args = build_parser().parse_args()
    print(args)
    num_devices = torch.cuda.device_count()
    print("using {} num_processes inference".format(num_devices))
    datas = json.load(open(args.json_path))
    print(len(datas))
    mp.set_start_method("spawn")
    chunks = P.split_list_into_chunks(datas, num_devices)
    procs = [mp.Process(target=inference, args=(args, r, chunks[r])) for r in range(num_devices)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join()
