"""CPU oracle for the PCDMs stage-2 inpainting denoising loop.

TEST INFRASTRUCTURE ONLY.  Nothing under ``pcdms_amd/`` may import this package; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it, and there
only as the checker (never as the thing measured or shipped).

PARITY UNPINNED: the arithmetic of this path lives in the un-vendored third-party package
``diffusers==0.24.0`` (pinned at /root/reference/README.md:37) which is neither under
/root/reference nor installable offline, and the reference holds no tests / golden vectors for
the path.  The oracle is therefore a plain-PyTorch fp32 restatement of the
published algorithm, anchored on the reference's own call sites:

* topology + forward order: src/models/stage2_inpaint_unet_2d_condition.py:66-448, :579-825
* conditioning assembly + denoise loop: src/pipelines/stage2_inpaint_pipeline.py:420-541
* DDIM configuration: pcdms_kaggle_demo.ipynb cell 15

It is pinned as far as is possible here by (a) the parameter count 868 876 804, (b) closed-form
scheduler known answers, (c) algebraic identities (Appendix C-6) and
(d) golden fixtures produced by running the reference's *own* ``forward`` / ``__call__`` on top of
the oracle's blocks (tests/golden/make_reference_wiring_fixtures.py).
"""
