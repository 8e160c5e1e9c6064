"""`utils.logger.Logger` as the unchanged training scripts use it
(/root/reference/trainer/train_transducer_bmuf_otfaug.py:19,57,130,135): running per-tag loss
averages written every `log_per_nframes` units, plus an overall summary."""
import time


class Logger(object):
    def __init__(self, log_file, log_per_nframes, tags, loss_per_frame=(1.0,)):
        self.log_file = log_file
        self.log_per_nframes = log_per_nframes
        self.tags = list(tags)
        n = len(self.tags)
        self.loss_per_frame = list(loss_per_frame) if len(loss_per_frame) == n else [1.0] * n
        self.num_frames = 0
        self.total_frames = 0
        self.loss = [0.0] * n
        self.total_loss = [0.0] * n
        self.start_time = self.log_time = time.time()

    def update_and_log(self, num_frames, loss):
        self.num_frames += num_frames
        self.total_frames += num_frames
        for i, v in enumerate(loss):
            self.loss[i] += v
            self.total_loss[i] += v
        if self.num_frames >= self.log_per_nframes:
            elapsed = time.time() - self.log_time
            for tag, v, per in zip(self.tags, self.loss, self.loss_per_frame):
                self.log_file.write('{}: {:.3f} \t'.format(tag, v / per / float(self.num_frames)))
            self.log_file.write('fps: {:.6f} k\n'.format(self.num_frames / elapsed / 1000))
            self.log_file.flush()
            self.num_frames = 0
            self.loss = [0.0] * len(self.tags)
            self.log_time = time.time()

    def summarize_and_log(self):
        for tag, v, per in zip(self.tags, self.total_loss, self.loss_per_frame):
            self.log_file.write('Finished, Overall Avg {}: {:.3f}\t'.format(
                tag, v / per / float(self.total_frames)))
        elapsed = time.time() - self.start_time
        self.log_file.write('Avg fps:{:.6f} k\n'.format(self.total_frames / elapsed / 1000))
        return self.total_loss[0], self.total_frames
