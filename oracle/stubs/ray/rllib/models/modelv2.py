class ModelV2:
    """Pass-through __call__: obs_flat = obs (already 2-D), then forward()."""

    def __init__(self, obs_space, action_space, num_outputs, model_config, name, framework):
        self.obs_space = obs_space
        self.action_space = action_space
        self.num_outputs = num_outputs
        self.model_config = model_config
        self.name = name or "default_model"
        self.framework = framework

    def __call__(self, input_dict, state=None, seq_lens=None):
        restored = dict(input_dict)
        restored["obs_flat"] = restored["obs"]
        outputs, state_out = self.forward(restored, state or [], seq_lens)
        return outputs, (state_out if state_out is not None else (state or []))
