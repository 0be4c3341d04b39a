def load_library():
    raise RuntimeError("libgm_hip not built yet")
