"""physicsvae_amd -- MI355X-native hot path of the PhysicsVAE supervised training loop.

    physicsvae_amd.torch_models        the reference's torch_models.py module API
    physicsvae_amd.train_physics_vae   the reference's trainer / CLI surface
    physicsvae_amd.model               PhysicsVAE module surface (state_dict layout kept)
    physicsvae_amd.engine              HipEngine: device arenas + C-ABI calls
    physicsvae_amd.csrc                hand-written gfx950 kernels + C ABI (include/pvae.h)

The compute path is libpvae_gfx950.so only; there is no CPU or eager-PyTorch fallback.
"""
__version__ = "0.1.0"
