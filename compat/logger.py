"""Drop-in for the reference's src/logger.py (logger.py:7-82): `from logger import Logger` (train_gan.py:13).

The reference pickles its Logger INSTANCE into every checkpoint (train_gan.py:112-118, resumed at :274), so a class
of this name under this module name must exist for `torch.load` to rebuild it -- and a Logger written here must carry
the attributes the reference's methods read (tracker / counter / mean / history / iterator / writer / log_path).

Same surface: safe(write), reset(), append(result, tag, n, mean), write(tag, metric_names), flush().  The
TensorBoard writer is optional (the `tensorboard` package is not in this image): without it safe(True) keeps
`writer = None` and the scalars go to stdout only -- which is what write() prints in the reference as well.
"""
from collections import defaultdict
from collections.abc import Iterable
from numbers import Number


def _summary_writer(path):
    try:
        from torch.utils.tensorboard import SummaryWriter
        return SummaryWriter(path)
    except Exception:                                   # tensorboard missing or unusable: log to stdout only
        return None


_FIELDS = ('tracker', 'counter', 'mean', 'iterator')


class Logger:
    def __init__(self, log_path):
        self.log_path = log_path
        self.writer = None
        self.history = defaultdict(list)
        for f in _FIELDS:
            setattr(self, f, defaultdict(int))

    # -- pickling: the writer (a file handle) never travels; a file written by another version of the class, or by the
    #    reference's own class, may miss fields -- they are rebuilt empty
    def __getstate__(self):
        state = dict(self.__dict__)
        state['writer'] = None
        return state

    def __setstate__(self, state):
        self.__init__(state.get('log_path', ''))
        for k, v in state.items():
            if k == 'writer':
                continue
            if k in _FIELDS and not isinstance(v, defaultdict):
                v = defaultdict(int, v)
            if k == 'history' and not isinstance(v, defaultdict):
                v = defaultdict(list, v)
            setattr(self, k, v)

    def safe(self, write):
        """safe(True) opens the epoch's writer; safe(False) closes it and files the epoch's means under history."""
        if write:
            self.writer = _summary_writer(self.log_path)
            return
        if self.writer is not None:
            self.writer.close()
            self.writer = None
        for name, value in self.mean.items():
            self.history[name].append(value)

    def reset(self):
        for f in ('tracker', 'counter', 'mean'):
            setattr(self, f, defaultdict(int))

    def append(self, result, tag, n=1, mean=True):
        """Track result[k] under '<tag>/<k>'; `mean`: fold it into the running mean weighted by n samples."""
        for k, value in result.items():
            name = f'{tag}/{k}'
            self.tracker[name] = value
            self.counter[name] += n
            if not mean:
                continue
            seen = self.counter[name]

            def fold(old, new):
                return ((seen - n) * old + n * new) / seen
            if isinstance(value, Number):
                self.mean[name] = fold(self.mean[name], value)
            elif isinstance(value, Iterable):
                value = list(value)
                old = self.mean[name] if name in self.mean else [0] * len(value)
                self.mean[name] = [fold(o, v) for o, v in zip(old, value)]
            else:
                raise ValueError('Not valid data type')

    def write(self, tag, metric_names):
        """Print '<info fields>  <metric>: <mean> ...' for the tag and mirror the scalars to the writer if there is one."""
        parts = []
        for k in metric_names:
            name = f'{tag}/{k}'
            m = self.mean[name]
            if isinstance(m, Number):
                parts.append(f'{k}: {m:.4f}')
                scalar = m
            elif isinstance(m, Iterable):
                m = tuple(m)
                parts.append(f'{k}: {m}')
                scalar = m[0]
            else:
                raise ValueError('Not valid data type')
            if self.writer is not None:
                self.iterator[name] += 1
                self.writer.add_scalar(name, scalar, self.iterator[name])
        info_name = f'{tag}/info'
        info = self.tracker[info_name]
        info = list(info) if isinstance(info, (list, tuple)) else []
        line = '  '.join(info[:2] + parts + info[2:])
        print(line)
        if self.writer is not None:
            self.iterator[info_name] += 1
            self.writer.add_text(info_name, line, self.iterator[info_name])

    def flush(self):
        if self.writer is not None:
            self.writer.flush()
