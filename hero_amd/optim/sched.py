"""Learning-rate schedules with the reference's formulas (optim/sched.py)."""


def noam_schedule(step, warmup_step=4000):
    return step / warmup_step if step <= warmup_step else (warmup_step ** 0.5) * (step ** -0.5)


def warmup_linear(step, warmup_step, tot_step):
    if step < warmup_step:
        return step / warmup_step
    return max(0, (tot_step - step) / (tot_step - warmup_step))


def get_lr_sched(global_step, opts):
    lr = opts.learning_rate * warmup_linear(global_step, opts.warmup_steps, opts.num_train_steps)
    return lr if lr > 0 else 1e-8
