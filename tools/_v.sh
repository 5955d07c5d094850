timeout 500 python tools/concurrent_clips.py 110 4 2>&1 | grep -E "side by side|rc=" | cut -c1-150
timeout 500 python tools/concurrent_clips.py 200 3 2>&1 | grep -E "side by side|rc=" | cut -c1-150
