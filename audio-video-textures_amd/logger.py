"""Tensorboard logger shim with the reference's method names (utils/logger.py:8-81).
Observability is outside the hot path: when tensorboardX is not installed every call is a no-op."""


class Logger:
    def __init__(self, log_dir, n_logged_samples=10, summary_writer=None):
        self._log_dir = log_dir
        self._w = None
        try:
            if summary_writer is None:
                from tensorboardX import SummaryWriter as summary_writer
            self._w = summary_writer(log_dir, flush_secs=1, max_queue=1)
        except Exception:
            print("Logger: tensorboardX unavailable, logging disabled ({})".format(log_dir))

    def log_scalar(self, scalar, name, step_):
        if self._w:
            self._w.add_scalar("{}".format(name), scalar, step_)

    def log_scalars(self, scalar_dict, group_name, step, phase):
        if self._w:
            self._w.add_scalars("{}_{}".format(group_name, phase), scalar_dict, step)

    def log_image(self, image, name, step):
        if self._w:
            self._w.add_image("{}".format(name), image, step)

    def log_figure(self, figure, name, step):
        if self._w:
            self._w.add_figure("{}".format(name), figure, step)

    def flush(self):
        if self._w:
            self._w.flush()
