def imread(path, *a, **k):
    import numpy as np
    from PIL import Image
    return np.asarray(Image.open(path))
