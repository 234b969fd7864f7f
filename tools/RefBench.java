// RefBench.java -- the reference itself as the CPU baseline (SURVEY.md 8(d): "preferred: the reference on a JVM").
//
// Calls only the reference's public API as its README shows it (README.md:63-79):
//     DefinitionReader.reader(File).read()  ->  Gorp;   gorp.extract(String)  ->  ExtractionResult or null
// T threads each run Gorp.extract over a contiguous shard of the same lines ("Gorp is fully thread-safe",
// core/Gorp.java:22), once to warm the JIT up and once timed.  Prints one JSON line.
//
// This repository ships the source only: bench.py compiles and runs it in a child process when `java`, `javac` and
// GORP_REFERENCE_CLASSPATH (gorp-core, dk.brics.automaton 1.11-8, jackson-jr 2.8.2) are all present on the box, and
// reports "reference JVM path: unavailable" otherwise -- as in this image, which has no JDK.
//
//     javac -cp "$GORP_REFERENCE_CLASSPATH" -d <dir> tools/RefBench.java
//     java  -cp "<dir>:$GORP_REFERENCE_CLASSPATH" RefBench <definition.grp> <lines.txt> <threads>
import java.io.File;
import java.nio.charset.StandardCharsets;
import java.nio.file.Files;
import java.util.List;
import java.util.concurrent.atomic.AtomicLong;

import com.salesforce.gorp.DefinitionReader;
import com.salesforce.gorp.ExtractionException;
import com.salesforce.gorp.ExtractionResult;
import com.salesforce.gorp.Gorp;

public class RefBench {
    public static void main(String[] args) throws Exception {
        final Gorp gorp = DefinitionReader.reader(new File(args[0])).read();
        // (ISO-8859-1: one char per byte, the code units the GPU path is given)
        final List<String> lines = Files.readAllLines(new File(args[1]).toPath(), StandardCharsets.ISO_8859_1);
        final int threads = Integer.parseInt(args[2]);
        final int n = lines.size();
        long bytes = 0;
        for (String s : lines) bytes += s.length();
        double seconds = 0;
        long matched = 0, failed = 0;
        for (int pass = 0; pass < 2; ++pass) {   // pass 0 warms the JIT up
            final AtomicLong m = new AtomicLong(), x = new AtomicLong();
            Thread[] pool = new Thread[threads];
            final long t0 = System.nanoTime();
            for (int t = 0; t < threads; ++t) {
                final int lo = (int) ((long) n * t / threads), hi = (int) ((long) n * (t + 1) / threads);
                pool[t] = new Thread(() -> {
                    long mm = 0, xx = 0;
                    for (int i = lo; i < hi; ++i) {
                        try {
                            ExtractionResult r = gorp.extract(lines.get(i));
                            if (r != null) ++mm;
                        } catch (ExtractionException e) {
                            ++xx;
                        }
                    }
                    m.addAndGet(mm);
                    x.addAndGet(xx);
                });
                pool[t].start();
            }
            for (Thread th : pool) th.join();
            seconds = (System.nanoTime() - t0) * 1e-9;
            matched = m.get();
            failed = x.get();
        }
        System.out.println(String.format(
            "{\"value\": %.1f, \"unit\": \"lines/s\", \"cores\": %d, \"kind\": \"reference\", \"lines\": %d, \"bytes\": %d, "
            + "\"seconds\": %.3f, \"matched\": %d, \"extraction_exceptions\": %d, \"java\": \"%s\"}",
            n / seconds, threads, n, bytes, seconds, matched, failed, System.getProperty("java.version")));
    }
}
