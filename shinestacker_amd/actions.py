"""Action layer of the drop-in boundary: StackJob, FocusStack, FocusStackBunch,
CombinedActions, SubAction.

Host-side mirror of the reference's job/action protocol for the hot path only
(8(b)).  Observable behaviour kept identical to the
reference:

* construction: StackJob(name, working_path, input_path='', callbacks=None) and
  job.add_action(a) assign ids and inherit callbacks (core/framework.py:149-172);
* paths: an action reads `input_path` (default: previous action's output) under the
  job's working_path and writes to `output_path` (default: its own name); an existing
  output directory is emptied when `scratch_output_dir` (stack_framework.py:57-99);
* output name "{output_dir}/{prefix}{first_input_name}" (stack.py:30-32), prefix 'stack_';
* callback sequence before_action -> step_counts -> after_step... -> save_plot ->
  after_action, `check_running` polled after each step, RunStopException on False
  (core/framework.py:90-103, :174-188, :222-229; stack.py:105-109);
* bunch partition get_bunches and the reference's step order -- run_step indexes
  chunks[count - 1], so the LAST bunch is fused first (stack.py:61-64, :94-97);
* CombinedActions frame/reference iteration order for step_process False/True
  (stack_framework.py:191-232, :246-302).

Not mirrored (out of scope): tqdm bars, coloured console logging, EXIF
copy, denoise, GUI hooks.  When the real `shinestacker` package is installed, its
own FocusStack/StackJob accept `shinestacker_amd.PyramidStack` directly.
"""
import logging
import os
import time

import numpy as np

from .defaults import constants
from .errors import BitDepthError, InvalidOptionError, RunStopException, ShapeError
from .imageio import read_img, write_img

_log = logging.getLogger("shinestacker_amd")


def _require_dir(path):
    if not os.path.exists(path):
        raise RuntimeError(f"Path does not exist: {path}")


def _join(base, rel):
    return base + ('' if base.endswith('/') else '/') + rel


class ActionBase:
    """Callbacks + messaging shared by jobs and actions (core/framework.py:65-146)."""

    def __init__(self, name, enabled=True):
        self.id = -1
        self.name = name
        self.enabled = enabled
        self.callbacks = None
        self.logger = None
        self.base_message = ''
        self._t0 = None

    def time(self):
        """core/framework.py:160-161: seconds since run() started"""
        return time.time() - self._t0

    def set_terminator(self, tqdm=False, end='\n'):
        """core/framework.py:112-116 switches the console handler's line terminator for its progress bars; there are no
        bars here (out of scope), the call is accepted and does nothing"""

    def callback(self, key, *args):
        cbs = getattr(self, 'callbacks', None)
        if cbs is not None:
            fn = cbs.get(key, None)
            if fn:
                return fn(*args)
        return None

    def get_logger(self):
        return self.logger if self.logger is not None else _log

    def print_message(self, msg='', level=logging.INFO, **_kw):
        self.base_message = self.name + (': ' + msg if msg != '' else '')
        self.get_logger().log(level, self.base_message)

    def sub_message(self, msg, level=logging.INFO, **_kw):
        self.get_logger().log(level, f"{self.base_message}{msg}")

    print_message_r = print_message
    sub_message_r = sub_message

    def run_core(self):
        pass

    def run(self):
        self._t0 = time.time()
        if not self.enabled:
            self.get_logger().warning(self.name + ": entire job disabled")
        self.callback('before_action', self.id, self.name)
        self.run_core()
        self.callback('after_action', self.id, self.name)
        self.get_logger().info(f"{self.name}: elapsed time: {time.time() - self._t0:.2f}s")
        self.get_logger().info(f"{self.name}: completed")


class StackJob(ActionBase):
    """stack_framework.py:13-24 + core/framework.py:149-188."""

    def __init__(self, name, working_path, input_path='', logger_name=None, callbacks=None,
                 **kwargs):
        _require_dir(working_path)
        self.working_path = working_path
        self.paths = [] if input_path == '' else [input_path]
        super().__init__(name, **kwargs)
        self.action_counter = 0
        self._actions = []
        if logger_name is not None:
            self.logger = logging.getLogger(logger_name)
        # 'tqdm' selects the reference's progress-bar callbacks (core/framework.py:158): bars are out of scope, the job runs
        self.callbacks = None if callbacks == 'tqdm' else callbacks

    def init(self, a):
        """stack_framework.py:23-24: the job hands itself to the action (paths, working directory)"""
        a.init(self)

    def add_action(self, a):
        a.id = self.action_counter
        self.action_counter += 1
        a.logger = self.logger
        a.callbacks = self.callbacks
        self.init(a)
        self._actions.append(a)

    def run_core(self):
        for a in self._actions:
            if not (a.enabled and self.enabled):
                what = " and ".join(w for w, off in (("action", not a.enabled),
                                                     ("job", not self.enabled)) if off)
                self.get_logger().warning(f"{a.name}: {what} disabled")
                continue
            if self.callback('check_running', self.id, self.name) is False:
                raise RunStopException(self.name)
            a.run()


class FrameDirectory:
    """Input/output directory plumbing (stack_framework.py:27-130)."""

    def __init__(self, name, input_path='', output_path='', working_path='',
                 plot_path=constants.DEFAULT_PLOTS_PATH, scratch_output_dir=True, resample=1,
                 reverse_order=constants.DEFAULT_FILE_REVERSE_ORDER, **_kwargs):
        self.name = name
        self.working_path = working_path
        self.plot_path = plot_path
        self.input_path = input_path
        self.output_path = output_path
        self.output_dir = None
        self.resample = resample
        self.reverse_order = reverse_order
        self.scratch_output_dir = scratch_output_dir
        self.input_full_path = None
        self.filenames = None

    def folder_list_str(self):
        """stack_framework.py:106-111"""
        return "folder: " + self.input_full_path.replace(self.working_path, '').lstrip('/')

    def folder_filelist(self):
        _dirpath, _, names = next(os.walk(self.input_full_path))
        files = sorted(n for n in names
                       if os.path.splitext(n)[-1][1:].lower() in constants.EXTENSIONS)
        if self.reverse_order:
            files.reverse()
        if self.resample > 1:
            files = files[0::self.resample]
        return files

    def set_filelist(self):
        self.filenames = self.folder_filelist()
        rel = self.input_full_path.replace(self.working_path, '').lstrip('/')
        self.print_message(f": {len(self.filenames)} files in folder: {rel}")

    def init_paths(self, job):
        if self.working_path == '':
            self.working_path = job.working_path
        _require_dir(self.working_path)
        if self.output_path == '':
            self.output_path = self.name
        self.output_dir = _join(self.working_path, self.output_path)
        if not os.path.exists(self.output_dir):
            os.makedirs(self.output_dir)
        else:
            existing = os.listdir(self.output_dir)
            if existing and self.scratch_output_dir and self.enabled:
                for fn in existing:
                    fp = os.path.join(self.output_dir, fn)
                    if os.path.isfile(fp):
                        os.remove(fp)
                self.print_message(f": output directory {self.output_path} content erased")
            elif existing and not self.scratch_output_dir:
                self.print_message(f": output directory {self.output_path} not empty, "
                                   "files may be overwritten or merged with existing ones.",
                                   level=logging.WARNING)
        if self.input_path == '':
            if len(job.paths) == 0:
                raise RuntimeError(f"Job {job.name} does not have any configured path")
            self.input_path = job.paths[-1]
        job.paths.append(self.output_path)
        self.input_full_path = _join(self.working_path, self.input_path)
        _require_dir(self.input_full_path)
        job.paths.append(self.output_path)


class StepList(ActionBase):
    """Counted step iteration (core/framework.py:191-229)."""

    def __init__(self, name, enabled=True):
        super().__init__(name, enabled)
        self.counts = None
        self.count = None

    def set_counts(self, counts):
        self.counts = counts
        self.callback('step_counts', self.id, self.name, self.counts)

    def begin(self):
        self.callback('begin_steps', self.id, self.name)

    def end(self):
        self.callback('end_steps', self.id, self.name)

    def run_step(self):
        pass

    def run_core(self):
        self.print_message('begin run')
        self.begin()
        self.count = 0
        while self.count < self.counts:
            self.run_step()
            self.count += 1
            self.callback('after_step', self.id, self.name, self.count)
            if self.callback('check_running', self.id, self.name) is False:
                raise RunStopException(self.name)
        self.end()


# ------------------------------------------------------------------------------ stacking
class _FocusStackCommon(FrameDirectory):
    def _init_stack(self, stack_algo, kwargs):
        self.stack_algo = stack_algo
        self.exif_path = kwargs.pop('exif_path', '')
        self.prefix = kwargs.pop('prefix', constants.DEFAULT_STACK_PREFIX)
        self.denoise_amount = kwargs.pop('denoise_amount', 0)
        self.plot_stack = kwargs.pop('plot_stack', constants.DEFAULT_PLOT_STACK)
        self.stack_algo.process = self
        self.frame_count = -1

    def focus_stack(self, filenames):
        """stack.py:26-52: fuse `filenames`, write the result, emit 'save_plot'."""
        self.sub_message_r(': reading input files')
        img_files = sorted(os.path.join(self.input_full_path, n) for n in filenames)
        stacked = self.stack_algo.focus_stack(img_files)
        parts = filenames[0].split(".")
        out_filename = f"{self.output_dir}/{self.prefix}{parts[0]}." + '.'.join(parts[1:])
        if self.denoise_amount > 0:
            raise InvalidOptionError("denoise_amount", self.denoise_amount,
                                     "post-stack denoise is outside the MI355X hot path")
        write_img(out_filename, stacked)
        if self.plot_stack:
            idx_str = f"{self.frame_count + 1:04d}" if self.frame_count >= 0 else ''
            title = f"{self.name}: {self.stack_algo.name()}"
            if idx_str != '':
                title += f"\nbunch: {idx_str}"
            self.callback('save_plot', self.id, title, out_filename)
        if self.frame_count >= 0:
            self.frame_count += 1
        return out_filename


class FocusStack(ActionBase, _FocusStackCommon):
    """stack.py:100-113."""

    def __init__(self, name, stack_algo, enabled=True, **kwargs):
        FrameDirectory.__init__(self, name, **kwargs)
        ActionBase.__init__(self, name, enabled)
        self._init_stack(stack_algo, kwargs)
        self.stack_algo.do_step_callback = True

    def init(self, job, _working_path=''):
        self.init_paths(job)

    def run_core(self):
        self.set_filelist()
        self.callback('step_counts', self.id, self.name,
                      self.stack_algo.steps_per_frame() * len(self.filenames))
        self.focus_stack(self.filenames)


def get_bunches(collection, n_frames, n_overlap):
    """stack.py:61-64."""
    step = n_frames - n_overlap
    return [collection[x:x + n_frames] for x in range(0, len(collection) - n_overlap, step)]


def shard_steps(n_steps, rank, world):
    """Contiguous block of step indices for `rank`: sizes differ by at most one, every step exactly once."""
    per, extra = divmod(n_steps, world)
    lo = rank * per + min(rank, extra)
    return range(lo, lo + per + (1 if rank < extra else 0))


class _Sharded:
    """Rank bookkeeping of the actions that split their independent steps over processes (one per GPU)."""
    rank, world = 0, 1

    def _shard_setup(self, shard, kwargs, device_holders=()):
        if shard == 'env':
            self.rank, self.world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
            for obj in device_holders:
                if hasattr(obj, "device"):
                    obj.device = int(os.environ.get("LOCAL_RANK", "0"))
        elif shard is not None:
            self.rank, self.world = int(shard[0]), int(shard[1])
        if not 0 <= self.rank < self.world:
            raise InvalidOptionError("shard", shard, "rank must be in [0, world)")
        if self.rank > 0:
            kwargs['scratch_output_dir'] = False   # only rank 0 empties the shared output directory

    def _marker(self):
        run = os.environ.get("TORCHELASTIC_RUN_ID") or os.environ.get("MASTER_PORT") or "0"
        return os.path.join(self.output_dir, f".shard_ready_{run}")

    def _wait_for_rank0(self, timeout=600.0):
        """Rank 0 has emptied the output directory (init_paths) before any rank writes into it."""
        if self.world == 1:
            return
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                dist.barrier()
                return
        except ImportError:
            pass
        import time
        if self.rank == 0:
            open(self._marker(), "w").close()
            return
        t0 = time.monotonic()
        while not os.path.exists(self._marker()):
            if time.monotonic() - t0 > timeout:
                raise RuntimeError(f"rank {self.rank}: rank 0 did not prepare {self.output_dir}")
            time.sleep(0.01)


class FocusStackBunch(StepList, _FocusStackCommon, _Sharded):
    """stack.py:67-97.

    Multi-GPU ("bunch mode: whole bunches per GPU, no collective"): bunches are independent
    stacks, so with ``shard=(rank, world)`` -- or ``shard='env'``, which reads RANK / WORLD_SIZE / LOCAL_RANK
    as torch.distributed.run sets them and puts the stacker on device LOCAL_RANK -- every process fuses a
    contiguous block of the bunch steps and writes those output files; file names, contents and the
    'bunch: NNNN' plot titles are the ones the single-process run produces.  Only rank 0 scratches the output
    directory; the other ranks wait for it (torch.distributed barrier when a process group exists, else a marker
    file), so that no rank writes into a directory that is still being emptied."""

    def __init__(self, name, stack_algo, enabled=True, shard=None, **kwargs):
        self._shard_setup(shard, kwargs, (stack_algo,))
        StepList.__init__(self, name, enabled)
        FrameDirectory.__init__(self, name, **kwargs)
        self._init_stack(stack_algo, kwargs)
        self._chunks = None
        self._steps = None
        self.frame_count = 0
        self.frames = kwargs.get('frames', constants.DEFAULT_FRAMES)
        self.overlap = kwargs.get('overlap', constants.DEFAULT_OVERLAP)
        self.stack_algo.do_step_callback = False
        if self.overlap >= self.frames:
            raise InvalidOptionError("overlap", self.overlap,
                                     "overlap must be smaller than batch size")

    def init(self, job, _working_path=''):
        self.init_paths(job)

    def begin(self):
        StepList.begin(self)
        self._chunks = get_bunches(self.folder_filelist(), self.frames, self.overlap)
        self._steps = list(shard_steps(len(self._chunks), self.rank, self.world))
        self.set_counts(len(self._steps))
        self._wait_for_rank0()

    def run_step(self):
        step = self._steps[self.count]
        self.frame_count = step   # the 'bunch: NNNN' title numbers bunches by their global step
        self.print_message_r(f"fusing bunch: {step + 1}/{len(self._chunks)}")
        # the reference indexes chunks[count - 1] with count starting at 0 (stack.py:97):
        # the last bunch goes first; every bunch is still fused exactly once.
        self.focus_stack(self._chunks[step - 1])


# ------------------------------------------------------------------------------ per-frame actions
class SubAction:
    """stack_framework.py:235-243."""

    def __init__(self, enabled=True):
        self.enabled = enabled

    def begin(self, process):
        pass

    def end(self):
        pass


class CombinedActions(StepList, FrameDirectory, _Sharded):
    """Per-frame sub-action pipeline with a reference frame
    (stack_framework.py:191-232 FramesRefActions + :246-302 CombinedActions)."""

    def __init__(self, name, actions=None, enabled=True, ref_idx=-1, step_process=False,
                 io_threads=2, shard=None, **kwargs):
        # SURVEY.md 8(e), alignment: without step_process every frame depends on the reference frame only, so
        # the frames split over processes like bunches do (shard=(rank, world) or 'env'; every rank reads the
        # reference frame itself, no collective); with step_process the frames form two serial chains
        self._actions = list(actions or [])
        self._shard_setup(shard, kwargs, self._actions)
        if self.world > 1 and step_process:
            raise InvalidOptionError("shard", shard, "step_process chains frames; they cannot be sharded")
        FrameDirectory.__init__(self, name, **kwargs)
        StepList.__init__(self, name, enabled)
        self.ref_idx = ref_idx
        self.step_process = step_process
        # codec work off the critical path (not in the reference, which reads, processes and writes one
        # file at a time): the next input file is decoded while the current frame is processed, and
        # output files are encoded in the background.  0 = strictly sequential.  With step_process the
        # next step reads the previous OUTPUT file, so writes stay synchronous there.
        self.io_threads = io_threads
        self._pool = None
        self._ahead = {}
        self._writes = []
        self.dtype = None
        self.shape = None
        self._idx = self._ref_idx = self._idx_step = None

    def init(self, job, _working_path=''):
        self.init_paths(job)

    def begin(self):
        StepList.begin(self)
        self.set_filelist()
        self._block = shard_steps(len(self.filenames), self.rank, self.world)
        self.set_counts(len(self._block))
        if self.ref_idx == -1:
            self.ref_idx = len(self.filenames) // 2
        for a in self._actions:
            if a.enabled:
                a.begin(self)
        self._wait_for_rank0()

    def end(self):
        self._drain_writes()
        if self._pool is not None:
            self._pool.shutdown(wait=True)
            self._pool = None
        self._ahead = {}
        for a in self._actions:
            if a.enabled:
                a.end()
        StepList.end(self)

    # -- background codec work
    def _io_pool(self):
        if self._pool is None and self.io_threads and self.io_threads > 0:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=int(self.io_threads))
        return self._pool

    def _drain_writes(self):
        writes, self._writes = self._writes, []
        for w in writes:
            w.result()   # re-raises an encoder / file-system error of a background write

    def _read_input(self, idx):
        fut = self._ahead.pop(idx, None)
        path = f"{self.input_full_path}/{self.filenames[idx]}"
        return fut.result() if fut is not None else read_img(path)

    def _decode_next(self, idx, idx_step):
        """Start decoding the file the next step will ask for (same order as run_step)."""
        pool = self._io_pool()
        if pool is None:
            return
        n = len(self.filenames)
        nxt = idx + idx_step
        if nxt == n:
            nxt = self.ref_idx - 1
        if self.world > 1 and nxt not in self._block:
            return
        if 0 <= nxt < n and nxt not in self._ahead and not (self.step_process and nxt == self.ref_idx):
            self._ahead[nxt] = pool.submit(read_img, f"{self.input_full_path}/{self.filenames[nxt]}")

    def img_ref(self, idx):
        base = self.output_dir if self.step_process else self.input_full_path
        img = read_img(f"{base}/{self.filenames[idx]}")
        if img is None:
            raise RuntimeError(f"Invalid file: {self.input_full_path}/{self.filenames[idx]}")
        self.dtype, self.shape = img.dtype, img.shape
        return img

    def run_step(self):
        n = len(self.filenames)
        if self.count == 0:
            self._idx = self.ref_idx if self.step_process else self._block.start
            self._ref_idx = self.ref_idx
            self._idx_step = +1
        self.print_message_r(f"step {self.count + 1}/{self.counts}: process file: "
                             f"{self.filenames[self._idx]}, reference: "
                             f"{self.filenames[self._ref_idx]}")
        self.run_frame(self._idx, self._ref_idx)
        if self._idx < n:
            if self.step_process:
                self._ref_idx = self._idx
            self._idx += self._idx_step
        if self._idx == n:
            self._idx = self.ref_idx - 1
            if self.step_process:
                self._ref_idx = self.ref_idx
            self._idx_step = -1

    def run_frame(self, idx, ref_idx):
        filename = self.filenames[idx]
        self.sub_message_r(': read input image')
        img = self._read_input(idx)
        self._decode_next(idx, self._idx_step if self._idx_step is not None else +1)
        if img is None:
            raise RuntimeError(f"Invalid file: {self.input_full_path}/{filename}")
        if self.dtype is not None and img.dtype != self.dtype:
            raise BitDepthError(self.dtype, img.dtype)
        if self.shape is not None and img.shape != self.shape:
            raise ShapeError(self.shape, img.shape)
        if len(self._actions) == 0:
            self.sub_message(": no actions specified.", level=logging.WARNING)
        for a in self._actions:
            if not a.enabled:
                self.get_logger().warning(f"{self.base_message}: sub-action disabled")
                continue
            if self.callback('check_running', self.id, self.name) is False:
                raise RunStopException(self.name)
            img = a.run_frame(idx, ref_idx, img)
        self.sub_message_r(': write output image')
        if img is not None:
            out = np.ascontiguousarray(img)
            pool = self._io_pool()
            if pool is None or self.step_process:
                write_img(self.output_dir + "/" + filename, out)
            else:
                if len(self._writes) >= 2 * int(self.io_threads):
                    self._writes.pop(0).result()
                self._writes.append(pool.submit(write_img, self.output_dir + "/" + filename, out))
        else:
            self.print_message("No output file resulted from processing input file: "
                               f"{self.input_full_path}/{filename}", level=logging.WARNING)
