"""Callback hooks with the reference's names (`multivae/trainers/base/callbacks.py:65-186`).  Only the
framework-independent callbacks are provided; wandb / mlflow / tensorboard writers are out of scope."""
import logging

logger = logging.getLogger(__name__)


class TrainingCallback:
    def on_init_end(self, training_config, **kwargs): pass
    def on_train_begin(self, training_config, **kwargs): pass
    def on_train_end(self, training_config, **kwargs): pass
    def on_epoch_begin(self, training_config, **kwargs): pass
    def on_epoch_end(self, training_config, **kwargs): pass
    def on_train_step_begin(self, training_config, **kwargs): pass
    def on_train_step_end(self, training_config, **kwargs): pass
    def on_eval_step_begin(self, training_config, **kwargs): pass
    def on_eval_step_end(self, training_config, **kwargs): pass
    def on_evaluate(self, training_config, **kwargs): pass
    def on_prediction_step(self, training_config, **kwargs): pass
    def on_save(self, training_config, **kwargs): pass
    def on_log(self, training_config, logs, **kwargs): pass


class CallbackHandler:
    def __init__(self, callbacks, model):
        self.callbacks = []
        for cb in callbacks:
            self.add_callback(cb)
        self.model = model

    def add_callback(self, callback):
        cb = callback() if isinstance(callback, type) else callback
        cb_class = callback if isinstance(callback, type) else callback.__class__
        if cb_class in [c.__class__ for c in self.callbacks]:
            logger.warning(f"You are adding a {cb_class} to the callbacks but there one is already used.")
        self.callbacks.append(cb)

    def call_event(self, event, training_config, **kwargs):
        for callback in self.callbacks:
            getattr(callback, event)(training_config, model=self.model, **kwargs)

    def on_log(self, training_config, logs, **kwargs):
        self.call_event("on_log", training_config, logs=logs, **kwargs)

    def __getattr__(self, name):
        if name.startswith("on_"):
            return lambda training_config=None, **kw: self.call_event(name, training_config, **kw)
        raise AttributeError(name)


class MetricConsolePrinterCallback(TrainingCallback):
    def on_log(self, training_config, logs, **kwargs):
        rank = kwargs.pop("rank", -1)
        if rank in (-1, 0):
            epoch_train_loss = logs.get("train_epoch_loss", None)
            epoch_eval_loss = logs.get("eval_epoch_loss", None)
            logger.info("--------------------------------------------------------------------------")
            if epoch_train_loss is not None:
                logger.info(f"Train loss: {epoch_train_loss:.4f}")
            if epoch_eval_loss is not None:
                logger.info(f"Eval loss: {epoch_eval_loss:.4f}")
