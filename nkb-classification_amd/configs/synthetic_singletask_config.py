# Same schema as the reference's configs/singletask_config.py (keys read by train.py / engine / factories);
# data comes from the in-memory synthetic source so the config runs anywhere a GPU is present.
device = "cuda:0"
enable_mixed_presicion = True      # bf16 compute, fp32 accumulate / statistics / master weights
enable_gradient_scaler = False     # bf16 needs no loss scaling
compile = False
log_gradients = False
show_full_current_loss_in_terminal = False
task = "single"
n_epochs = 2
backbone_state_policy = {0: "unfreeze"}
classes = [str(i) for i in range(10)]
train_data = {"type": "synthetic", "n_images": 256, "classes": classes, "batch_size": 64, "shuffle": True, "num_workers": 0}
val_data = {"type": "synthetic", "n_images": 128, "classes": classes, "batch_size": 64, "shuffle": False, "num_workers": 0, "seed": 4321}
train_pipeline = None
val_pipeline = None
model = {"task": task, "model": "resnet18", "pretrained": False, "backbone_dropout": 0.0, "classifier_dropout": 0.0,
         "classifier_initialization": "kaiming_normal_"}
optimizer = {"type": "nadam", "lr": 1e-4, "weight_decay": 0.2, "backbone_lr": 1e-4, "backbone_weight_decay": 0.01,
             "classifier_lr": 1e-3, "classifier_weight_decay": 0.2}
lr_policy = {"type": "cosine", "n_epochs": n_epochs}
criterion = {"task": task, "type": "CrossEntropyLoss"}
experiment = {"comet": None, "local": {"path": "runs/synthetic_single"}}
