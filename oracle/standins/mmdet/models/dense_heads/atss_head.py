def reduce_mean(tensor):
    return tensor
