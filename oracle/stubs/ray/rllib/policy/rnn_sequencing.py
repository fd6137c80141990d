def add_time_dimension(*args, **kwargs):
    raise NotImplementedError("LSTM path is outside the PhysicsVAE training path")
