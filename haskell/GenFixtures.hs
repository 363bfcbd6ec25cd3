{-# LANGUAGE ScopedTypeVariables #-}

{-| GenFixtures -- outputs of the REFERENCE's own Haskell on fixed inputs, for the two rows of the hot path that only a
    GHC build can pin: @fmDemod@ (A5: Demod.hs:21-46 on top of GHC base's @Data.Complex@ and the @RealFloat@ default
    @atan2@) and the Pipes' state machines (A9: @firDecimator@ / @firResampler@ / @firFilter@, Filter.hs:532-727, with the
    sequential cross-buffer kernels FilterInternal.hs:397-423).

    NOT COMPILED ANYWHERE IN THIS REPOSITORY'S BUILD IMAGE (no GHC there).  A maintainer of adamwalker/sdr runs it once:

    > python tests/golden/make_haskell_inputs.py /tmp/hsfix          # this repository: writes the inputs (raw little-endian)
    > # in a checkout of adamwalker/sdr: add to sdr.cabal
    > #   executable gen-fixtures
    > #     main-is: GenFixtures.hs      (this file)    build-depends: base, sdr, vector, pipes, filepath
    > cabal run gen-fixtures -- /tmp/hsfix                            # writes /tmp/hsfix/*_out.*
    > python tests/golden/pack_haskell.py /tmp/hsfix                  # this repository: -> tests/golden/haskell_fixtures.npz
    > python -m pytest tests/test_golden_haskell.py                   # oracle (CPU) and libsdr_hip.so (GPU) against them

    Every kernel is the AVX variant (what @featureSelect@ picks on any AVX host, CPUID.hs:91-92), every Pipe runs with
    @blockSizeOut = 8192@ on 8192-element input blocks as examples/fm/fm.hs:17,34-41 does; a second fmDemod run uses ragged
    blocks.  Files are raw arrays: @.u8@ bytes, @.f32@ floats, @.cf32@ interleaved (re, im) floats.
-}
module Main (main) where

import           Control.Monad         (forM_, when)
import           Data.Complex
import           Data.IORef
import           Foreign.C.Types       (CUChar)
import           Foreign.ForeignPtr    (mallocForeignPtrArray, withForeignPtr)
import           Foreign.Storable      (Storable, sizeOf)
import qualified Data.Vector.Generic   as VG
import qualified Data.Vector.Storable  as VS
import           Pipes
import qualified Pipes.Prelude         as P
import           System.Environment    (getArgs)
import           System.FilePath       ((</>))
import           System.IO

import           SDR.Demod
import           SDR.Filter
import           SDR.Util

blockSize :: Int
blockSize = 8192

readVec :: forall a. Storable a => FilePath -> IO (VS.Vector a)
readVec path = withBinaryFile path ReadMode $ \h -> do
    bytes <- fromIntegral <$> hFileSize h
    let sz = sizeOf (undefined :: a)
        n  = bytes `div` sz
    fp  <- mallocForeignPtrArray n
    got <- withForeignPtr fp $ \p -> hGetBuf h p (n * sz)
    when (got /= n * sz) $ error ("short read: " ++ path)
    return (VS.unsafeFromForeignPtr0 fp n)

writeVecs :: forall a. Storable a => FilePath -> [VS.Vector a] -> IO ()
writeVecs path vs = withBinaryFile path WriteMode $ \h ->
    forM_ vs $ \v -> VS.unsafeWith v $ \p -> hPutBuf h p (VS.length v * sizeOf (undefined :: a))

-- | whole blocks only: the Python side feeds the restated Pipes the same list
blocksOf :: Storable a => Int -> VS.Vector a -> [VS.Vector a]
blocksOf n v
    | VS.length v < n = []
    | otherwise       = VS.take n v : blocksOf n (VS.drop n v)

-- | blocks of the given sizes, then whole 8192-blocks of what is left
raggedBlocks :: Storable a => [Int] -> VS.Vector a -> [VS.Vector a]
raggedBlocks (s : ss) v | VS.length v >= s = VS.take s v : raggedBlocks ss (VS.drop s v)
raggedBlocks _ v = blocksOf blockSize v

-- | every vector the Pipe yields for these inputs, in order
runPipe :: [a] -> Pipe a b IO () -> IO [b]
runPipe inputs pipe = do
    ref <- newIORef []
    runEffect $ each inputs >-> pipe >-> P.mapM_ (\x -> modifyIORef' ref (x :))
    reverse <$> readIORef ref

floats :: FilePath -> IO [Float]
floats path = VS.toList <$> (readVec path :: IO (VS.Vector Float))

main :: IO ()
main = do
    args <- getArgs
    let dir = case args of
                  (d : _) -> d
                  _       -> error "usage: gen-fixtures <directory written by tests/golden/make_haskell_inputs.py>"

    -- A5: fmDemod (Demod.hs:40-46), 8192-blocks and ragged blocks of the same stream
    iq :: VS.Vector (Complex Float) <- readVec (dir </> "demod_in.cf32")
    y1 <- runPipe (blocksOf blockSize iq) fmDemod
    writeVecs (dir </> "demod_out.f32") (y1 :: [VS.Vector Float])
    y2 <- runPipe (raggedBlocks [1, 2, 777, 4096, 129, 8191] iq) fmDemod
    writeVecs (dir </> "demod_ragged_out.f32") (y2 :: [VS.Vector Float])

    -- A2 + A6 + A9: firDecimator /8 (fastDecimatorAVXC = mkDecimatorC 4 decimateCAVXRC, Filter.hs:346-349)
    tapsD <- floats (dir </> "taps_decim.f32")
    deci  <- fastDecimatorAVXC 8 tapsD
    xd :: VS.Vector (Complex Float) <- readVec (dir </> "decim_in.cf32")
    d  <- runPipe (blocksOf blockSize xd) (firDecimator deci blockSize)
    writeVecs (dir </> "decim_out.cf32") d

    -- A3 + A8 + A9: firResampler 3/10 (fastResamplerAVXR = mkResampler 8 resampleCAVXRR, Filter.hs:461-465)
    tapsR <- floats (dir </> "taps_resamp.f32")
    resp  <- fastResamplerAVXR 3 10 tapsR
    xr :: VS.Vector Float <- readVec (dir </> "resamp_in.f32")
    z  <- runPipe (blocksOf blockSize xr) (firResampler resp blockSize)
    writeVecs (dir </> "resamp_out.f32") z

    -- A4 + A7 + A9: firFilter, symmetric (fastFilterSymAVXR = mkFilterSymR filterCAVXSymmetricRR, Filter.hs:253-255)
    half <- floats (dir </> "taps_audio_half.f32")
    filt <- fastFilterSymAVXR half
    xf :: VS.Vector Float <- readVec (dir </> "filt_in.f32")
    a  <- runPipe (blocksOf blockSize xf) (firFilter filt blockSize)
    writeVecs (dir </> "filt_out.f32") a

    -- the whole receiver, examples/fm/fm.hs:34-41 (fresh records: the ones above carry no state, but the resampler's startDat does)
    deci' <- fastDecimatorAVXC 8 tapsD
    resp' <- fastResamplerAVXR 3 10 tapsR
    filt' <- fastFilterSymAVXR half
    u8 :: VS.Vector CUChar <- readVec (dir </> "rx_in.u8")
    audio <- runPipe (blocksOf (2 * blockSize) u8) $
                 P.map interleavedIQUnsignedByteToFloatAVX
             >-> firDecimator deci' blockSize
             >-> fmDemod
             >-> firResampler resp' blockSize
             >-> firFilter filt' blockSize
             >-> P.map (VG.map (* 0.2))
    writeVecs (dir </> "rx_out.f32") (audio :: [VS.Vector Float])
    putStrLn ("wrote demod_out.f32 demod_ragged_out.f32 decim_out.cf32 resamp_out.f32 filt_out.f32 rx_out.f32 in " ++ dir)
