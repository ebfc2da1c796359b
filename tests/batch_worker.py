"""One rank of the world_size-N rehearsal of the sharded batch driver (no GPU: the per-scan work is a stub)."""
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "scripts")]


def main():
    rank, world, port, root, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], Path(sys.argv[4]), Path(sys.argv[5])
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import run_batch

    def stub(cfg):
        name = cfg.paths.recon_path.parent.parent.name
        if name == "broken":
            raise RuntimeError("boom")
        (out / f"{name}.ran_on").write_text(str(rank))
        time.sleep(0.01)

    rows = run_batch.main(run_batch.BatchConfig(root, out), run_scan=stub)
    (out / f"report.rank{rank}.json").write_text(json.dumps(rows))
    print("ok")


if __name__ == "__main__":
    main()
