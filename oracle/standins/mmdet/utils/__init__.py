import logging


def get_root_logger(log_file=None, log_level=logging.INFO):
    return logging.getLogger('mmdet')
