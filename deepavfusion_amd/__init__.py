"""MI355X-native (gfx950) implementation of the DeepAVFusion / AVMAE pre-training hot path.

Module paths mirror the reference (stoneMo/DeepAVFusion) so that its ``train.py`` flow is a
drop-in: ``deepavfusion_amd.models.{vits,fusion_blocks,deepavfusion,avmae}`` and
``deepavfusion_amd.util.{pos_embed,lr_sched,misc,distributed}``.  All device work is done by the
hand-written HIP kernels in ``csrc/`` (``libdavfusion_hip.so``) through the C ABI of
``include/dav_kernels.h``; there is no PyTorch/CPU fallback for the compute path.
"""
__version__ = '0.1.0'
