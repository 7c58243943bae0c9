"""pcdms_amd -- the PCDMs stage-2 pose-conditioned inpainting denoise loop, MI355X (gfx950) native.

Drop-in objects for the reference's ``pipe.unet`` / ``pipe.scheduler``
(/root/reference/stage2_batchtest_inpaint_model.py:125-132) backed by hand-written HIP kernels in
``pcdms_amd/lib/libpcdm.so`` (C-ABI: include/pcdm.h).  See DESIGN.md / INTEGRATION.md.
"""
from .parallel import run_sharded, split_list_into_chunks  # noqa: F401
from .pipeline import (PCDMsPipeline, Simple_Stage2_InpaintDiffusionPipeline, Stage2_InpaintDiffusionPipeline,  # noqa: F401
                       Stage2_InpaintDiffusionPipelineOutput, Stage3_RefinedDiffusionPipeline)
from .schedulers import DDIMScheduler, DDPMScheduler, UnCLIPScheduler, UniPCMultistepScheduler  # noqa: F401
from .unet import (Stage2_InapintUNet2DConditionModel, Stage2InpaintUNet, UNet2DConditionModel,  # noqa: F401
                   UNet2DConditionOutput)

from .vae import AutoencoderKL  # noqa: F401,E402
from .cond import ControlNetConditioningEmbedding, ImageProjModel_p  # noqa: F401,E402
from .prior import Stage1_PriorPipeline, Stage1_PriorTransformer  # noqa: F401,E402
from .encoders import CLIPVisionModelWithProjection, Dinov2Model  # noqa: F401,E402

__version__ = "0.1.0"
